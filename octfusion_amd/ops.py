"""Tensor-level wrappers over the C ABI (include/ofx.h).

Every function here launches hand-written HIP kernels from libofx.so on the
current torch HIP stream.  Inputs must be CUDA (HIP) tensors; nothing here has
a CPU implementation -- a CPU tensor raises.
"""
import contextlib
import os

import torch

from . import _lib
from ._lib import call, ptr, stream

ACT = {None: 0, 'none': 0, 'silu': 1, 'gelu': 2}

# bench.py sets this to a list to time every fused-GraphConv launch with HIP events on the
# launching stream: entries are (start_event, end_event, algorithmic_flops, algorithmic_bytes).
GRAPHCONV_PROFILE = None


def _meta(kind, flops=0.0, nbytes=0.0, tag=None):
    """Label the NEXT _lib.call for bench.py's per-class tail accounting (_lib.PROFILE): class, ALGORITHMIC flops and
    bytes of the launch (what the operator has to compute / move, not what the kernel happens to), a shape tag."""
    if _lib.PROFILE is not None:
        _lib.META = (kind, float(flops), float(nbytes), tag)


def _chk(t, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.OfxError('octfusion_amd ops need HIP tensors (no CPU path)')
    if t.dtype != dtype:
        raise TypeError('expected %s, got %s' % (dtype, t.dtype))


def _row_major(t):
    """(tensor, leading dimension) for a 2-D fp32 tensor whose rows are contiguous."""
    _chk(t)
    if t.dim() != 2:
        raise ValueError('expected a 2-D tensor')
    if t.stride(1) != 1 and t.shape[1] != 1:
        t = t.contiguous()
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    if ld < t.shape[1]:
        t = t.contiguous()
        ld = t.shape[1]
    return t, ld


_WS = {}


PRECISIONS = {'bf16x3': 0, 'fp32': 1, 'fp16': 2, 'fp16x3': 3}
DEFAULT_PRECISION = 'fp16x3'


def set_precision(mode):
    """'fp16x3' (default): operands as fp16 hi + lo pairs, three fp16 MFMAs per product, fp32 accumulate -- the fp32
    reference's rounding class (~2^-21 per product) on the 16-bit matrix pipe;
    'bf16x3': the same scheme with bf16 pairs (2^-16 per product; rounds 1-2's default, kept for A/B);
    'fp32' (exact fp32 MFMA); 'fp16' (reduced precision: single-pass fp16 MFMA in the planes GraphConv, every other
    contraction in bf16 pairs)."""
    call('ofx_set_precision', PRECISIONS[mode])


def get_precision():
    code = _lib.lib().ofx_get_precision()
    return [k for k, v in PRECISIONS.items() if v == code][0]


# Precision policy: parts of a step that run in another mode than the global one WHEN the global mode is 'bf16x3'
# (an A/B instrument of the precision study, profiles/r03/precision_attribution.json: with bf16 pairs the dense lr net
# is where a stand-alone lr step loses its accuracy -- element-wise p99.9 8.9e-3 -> 4.9e-4 with that net in exact fp32
# -- while the hr step's error sits in the wide GraphConvs).  The default mode 'fp16x3' needs no such help and ignores
# the policy.  Keys: 'dense_net' (the dense 16^3 U-Net as a stand-alone stage), 'small_gemm' (GEMMs / register-staged
# GraphConvs with <= 64 channels).
POLICY = {'dense_net': None, 'small_gemm': None}


@contextlib.contextmanager
def policy_scope(part):
    """Run the enclosed launches in POLICY[part] when the global mode is 'bf16x3' (no-op in every other mode and while
    POLICY[part] is None, the shipped setting)."""
    want = POLICY.get(part)
    L = _lib.lib()
    if want is None or L.ofx_get_precision() != PRECISIONS['bf16x3'] or want == 'bf16x3':
        yield
        return
    call('ofx_set_precision', PRECISIONS[want])
    try:
        yield
    finally:
        call('ofx_set_precision', PRECISIONS['bf16x3'])


# ---- operand planes (csrc/ofx_gemm2.hip): the LDS-DMA GraphConv and its producers --------------------
PLANES_ATTR = '_ofx_planes'
USE_PLANES = True            # A/B switch: False keeps every GraphConv on the register-staged kernel
# 256 x 128 tiles of a layer below which the register-staged kernel (128 x 128 tiles, split-K) is used instead of the
# planes kernel.  128 until round 3; the persistent stream-K launch balances any tile count over the chip, which moved
# the break-even down: a ONE-shape hr step (the generate regime) 3.66 -> 3.21 ms at 32, nothing more below
# (profiles/r03/generate_b1_min_tiles.txt; OFX_PLANES_MIN_TILES: A/B knob of tools/generate_probe.py)
PLANES_MIN_TILES = int(os.environ.get('OFX_PLANES_MIN_TILES', '32'))


def planes_mode():
    """Planes format that goes with the contraction precision: 3 = fp16 pairs (fp16x3), 2 = bf16 pairs (bf16x3),
    1 = single fp16 (reduced precision), 0 = none (exact fp32, or USE_PLANES off)."""
    if not USE_PLANES:
        return 0
    code = _lib.lib().ofx_get_precision()
    return {0: 2, 1: 0, 2: 1, 3: 3}[code]


def planes_pairs(mode):
    """modes 2 (bf16) / 3 (fp16): a 128-B line is [hi x 32 | lo x 32] of a 32-channel chunk -- the bytes of the fp32
    row, so the planes alias an fp32-shaped buffer; mode 1: fp16 row-major, 64 channels per line."""
    return mode in (2, 3)


def planes_of(t):
    return getattr(t, PLANES_ATTR, 0)


def _planes_chunk(mode):
    return 32 if planes_pairs(mode) else 64


def _planes_out(n, C, mode, device, out=None):
    """(tensor, row pitch in bytes) for a planes tensor of n rows x C channels."""
    if planes_pairs(mode):
        if out is None:
            out = torch.empty(n, C, dtype=torch.float32, device=device)
        assert out.dtype == torch.float32 and out.shape == (n, C) and out.stride(1) == 1
        ld = out.stride(0) * 4 if n > 1 else C * 4
    else:
        out = torch.empty(n, C, dtype=torch.float16, device=device)
        ld = C * 2
    assert out.data_ptr() % 128 == 0 and ld % 128 == 0, 'planes need 128-B aligned rows'
    return out, ld


def planes_ok(t, mode):
    """can tensor `t` (fp32 [n, C]) be overwritten in place by its hi / lo pair planes (modes 2, 3)?"""
    return (planes_pairs(mode) and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and
            t.data_ptr() % 128 == 0 and (t.stride(0) * 4) % 128 == 0 and t.shape[1] % 32 == 0)


def planes_split(x, mode, Cpad=None, out=None):
    """fp32 [n, C] -> planes (C zero-padded to Cpad)."""
    x, ldx = _row_major(x)
    n, C = x.shape
    ch = _planes_chunk(mode)
    Cpad = Cpad if Cpad is not None else (C + ch - 1) // ch * ch
    out, ldo = _planes_out(n, Cpad, mode, x.device, out)
    range_words(x.device)
    _meta('planes_split', 0, 4.0 * n * C + (4.0 if planes_pairs(mode) else 2.0) * n * Cpad, (n, C, Cpad))
    call('ofx_planes_split', ptr(x), ldx, n, C, Cpad, mode, ptr(out), ldo, stream())
    setattr(out, PLANES_ATTR, mode)
    return out


def planes_merge(p, mode=None):
    """planes -> fp32 [n, C] (tests)."""
    mode = mode or planes_of(p)
    n, C = p.shape
    ld = p.stride(0) * (4 if planes_pairs(mode) else 2)
    out = torch.empty(n, C, dtype=torch.float32, device=p.device)
    call('ofx_planes_merge', ptr(p), ld, n, C, mode, ptr(out), C, stream())
    return out


class PackedPlanes:
    """GraphConv weights packed as [k tile][cout][128-B line] for the planes kernel."""

    def __init__(self):
        self.t = None
        self.key = None

    def get(self, w, cin, nt, mode, cin_pad=None):
        """cin_pad > cin: the layer's input is zero-padded to cin_pad channels (weights get zero rows)."""
        cin_pad = cin if cin_pad is None else cin_pad
        key = (w.data_ptr(), w._version, tuple(w.shape), cin, nt, mode, cin_pad)
        if key == self.key:
            return self
        _chk(w)
        w = w.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        K, N = w.shape
        ntc = nt if nt > 1 else 0
        assert K == 7 * (cin + ntc)
        k_logical, cin_logical = K, cin                      # what the algorithmic flop / byte counts use
        if cin_pad != cin:
            wp = torch.zeros(7, cin_pad + ntc, N, dtype=torch.float32, device=w.device)
            w3 = w.view(7, cin + ntc, N)
            wp[:, :cin] = w3[:, :cin]
            wp[:, cin_pad:] = w3[:, cin:]
            w = wp.view(7 * (cin_pad + ntc), N)
            cin, K = cin_pad, 7 * (cin_pad + ntc)
        L = _lib.lib()
        out = torch.empty(L.ofx_planes_packed_bytes(cin, nt, N, mode), dtype=torch.uint8, device=w.device)
        call('ofx_pack_weights_planes', ptr(w), N, 1, cin, nt, N, mode, ptr(out), stream())
        self.t, self.key, self.N, self.K, self.cin, self.nt = out, key, N, k_logical, cin_logical, nt
        self.nkt = L.ofx_planes_packed_ktiles(cin, nt, mode)
        return self


def graphconv_planes(xp, mode, seg_ptr, col, ext, pw, cin, nt, tf_planes=None, bias=None, emb=None, batch_id=None,
                     res=None, out=None, stats=None):
    """Fused GraphConv on operand planes (ofx_graphconv_fwd_planes)."""
    assert planes_of(xp) == mode and xp.shape[1] == cin
    N = xp.shape[0]
    bpc = 4 if planes_pairs(mode) else 2
    ldx = xp.stride(0) * bpc if N > 1 else cin * bpc
    if out is None:
        out = torch.empty(N, pw.N, dtype=torch.float32, device=xp.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    nbr_ext, multi_seg, n_multi = ext
    aux = getattr(xp, AUX_ATTR, None)           # written by the producer of xp (group_norm(..., aux_graph=))
    aux_ready = 1 if aux is not None else 0
    if aux is None:
        aux = torch.empty((n_multi + 1) * ldx, dtype=torch.uint8, device=xp.device)
    assert aux.numel() >= (n_multi + 1) * ldx
    lde = ldr = ldt = 0
    if emb is not None:
        emb, lde = _row_major(emb)
    if emb is not None or stats is not None:
        _chk(batch_id, torch.int32)
    if res is not None:
        res, ldr = _row_major(res)
    if tf_planes is not None:
        ldt = tf_planes.stride(0) * bpc
    _chk(bias)
    _chk(stats, torch.float64)
    prof = GRAPHCONV_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(xp.device)
    sync = sync_words(xp.device)
    _meta('graphconv', 2.0 * N * pw.K * pw.N, 0, (N, pw.cin, pw.N, 'planes', 'emb' if emb is not None else '',
                                                  'res' if res is not None else '', 'stats' if stats is not None else ''))
    call('ofx_graphconv_fwd_planes', ptr(xp), ldx, cin, N, ptr(seg_ptr), ptr(col), ptr(nbr_ext), ptr(multi_seg),
         n_multi, ptr(aux), aux.numel() * aux.element_size(), ptr(tf_planes), ldt, nt, ptr(pw.t), pw.N, ptr(bias),
         ptr(emb), lde, ptr(batch_id) if (emb is not None or stats is not None) else None, ptr(res), ldr, ptr(out),
         ldc, ptr(stats), pw.N, ptr(ws), ws.numel(), ptr(sync), sync.numel() * 4, mode, aux_ready, stream())
    if prof is not None:
        e1.record()
        E = col.numel()
        s_in = 4.0 if planes_pairs(mode) else 2.0
        flops = 2.0 * N * pw.K * pw.N
        nbytes = s_in * (E * pw.cin + pw.K * pw.N) + 4.0 * N * pw.N + 8.0 * E
        prof.append((e0, e1, flops, nbytes, pw.N, ('graph2' if planes_pairs(mode) else 'graph2h', N, pw.cin, pw.N)))
    return out


# OFX_SIDE_STREAM=1: run the 1x1 skip convolutions of the res-blocks on a second stream, concurrently with the conv1
# chain.  Off by default: measured SLOWER on MI355X (hr step 9.44 ms vs 9.19 ms, graph replay) -- the dense GEMM's
# blocks take CU slots from the co-resident GraphConv blocks instead of only filling its tile-count tail.
SIDE_STREAM = os.environ.get('OFX_SIDE_STREAM', '0') == '1'
_SIDE = {}


def set_lane_cus(cus):
    """Compute units a persistent GraphConv launch is planned for while sampler.sample_loop runs a stage as lanes
    (ofx_set_gconv_cus, include/ofx.h); 0 = the whole device."""
    call('ofx_set_gconv_cus', int(cus))


def side_stream(device):
    """The per-device side stream for work that is independent of the main chain (joined before its result is used)."""
    s = _SIDE.get(device.index)
    if s is None:
        s = _SIDE[device.index] = torch.cuda.Stream(device)
    return s


# OFX_FORK=1 (default 0: measured slower inside a replayed hipGraph): work that does not depend on the activation chain -- the embedding MLPs of both nets, the 1x1
# residual convolutions of the dense net's ResnetBlocks -- is issued on a second stream and joined where its result
# is consumed.  Inside a captured step these become parallel branches of the hipGraph: ~19 launches of 5-15 us leave
# the step's critical path.  (The sparse net's 1x1 skip convolutions stay on the main stream: next to a persistent
# GraphConv launch that owns every CU they were measured slower, OFX_SIDE_STREAM above.)
FORK = int(os.environ.get('OFX_FORK', '0'))      # 0: off; 1: the embedding chains; 2: + the dense net's residual 1x1 convolutions
_FORK = {}


def fork_stream(device, level=1):
    """(main, fork) streams of `device` when forking is on (FORK >= level) and the tensor lives on a HIP device, else
    (None, None).  The fork stream is NOT sampler.sample_loop's warm-up / capture stream (side_stream): a fork happens
    inside a step.
    Allocator invariant: tensors produced on the fork stream are consumed on `main` without
    `record_stream`.  That is safe only because EVERY fork begins with `fork.wait_stream(main)` (below) and callers join
    with `main.wait_stream(fork)` before the result is read: a block the caching allocator hands out again on the fork
    stream cannot be written before main has passed the wait of the next fork, i.e. after main has finished reading the
    previous result.  Do not allocate on the fork stream outside a fork_stream() ... join pair."""
    if FORK < level or device.type != 'cuda':
        return None, None
    s = _FORK.get(device.index)
    if s is None:
        s = _FORK[device.index] = torch.cuda.Stream(device)
    main = torch.cuda.current_stream(device)
    if main.cuda_stream == s.cuda_stream:
        return None, None
    s.wait_stream(main)
    return main, s


_SYNC = {}
SYNC_WORDS = 4096


def _stream_id(device):
    return torch.cuda.current_stream(device).cuda_stream if device.type == 'cuda' else 0


def sync_words(device):
    """Flag words of the persistent stream-K GraphConv (include/ofx.h), one buffer per (device, STREAM): launches on
    one stream are ordered and may share them, launches on different streams (or graphs captured on different streams)
    run concurrently and must not.  Zero at allocation, every launch leaves them zero; the LAST word is the sticky
    error flag (sync_error())."""
    key = (device.type, device.index, _stream_id(device))
    t = _SYNC.get(key)
    if t is None:
        _no_capture('the flag words')
        t = _SYNC[key] = torch.zeros(SYNC_WORDS, dtype=torch.int32, device=device)
    return t


def _no_capture(what):
    """Per-stream scratch must exist BEFORE a capture on that stream: allocated inside one, the zero-fill of
    the flag words becomes a graph node that re-zeroes the sticky error word on every replay, and both buffers land in the
    graph's private pool.  sampler.sample_loop / bench.py warm up and capture on the SAME stream, so this never fires
    there; a caller that captures on a stream of its own must run one eager step on it first."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise _lib.OfxError('%s of this stream would be allocated inside a hipGraph capture: run one eager step on the '
                            'capture stream first (sampler.sample_loop does)' % what)


def sync_error(device):
    """True if a bounded wait of a persistent launch on `device` ever gave up, or a launch left a flag set (host sync)."""
    return any(int(t.abs().sum().item()) != 0 for k, t in _SYNC.items() if k[:2] == (device.type, device.index))


def raise_on_sync_error(device):
    """Production check, once per sampling call (sampler.sample_loop, generate, bench): a bounded wait of the persistent
    GraphConv launch gave up -> the results since then are invalid.  Clears the words, switches this process to the
    one-tile-per-block launch (no inter-block waits) and raises."""
    if not _SYNC or not sync_error(device):
        return
    for k, t in _SYNC.items():
        if k[:2] == (device.type, device.index):
            t.zero_()
    call('ofx_set_gconv_persistent', 0)
    raise _lib.OfxError('a flag wait of the persistent GraphConv launch gave up on %s: the results of this call are '
                        'invalid; the process now uses the one-tile-per-block launch (ofx_set_gconv_persistent(0))' % device)


_RANGE = {}
AUTO_RANGE_FALLBACK = True     # raise_on_range_error switches the process to 'bf16x3' before raising (the retry then fits)


class OfxRangeError(_lib.OfxError):
    """An operand left the range the fp16-pair contraction covers (include/ofx.h, fp16x3 range guard)."""


def range_words(device):
    """The device's sticky fp16x3 range-guard words (include/ofx.h), registered with the library on first use:
    [0] = operands ofx_planes_split found outside the fp16 range (|x| > 65504 or non-finite)."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _RANGE.get(key)
    if t is None:
        with torch.cuda.device(key[1]):
            t = _RANGE[key] = torch.zeros(4, dtype=torch.int32, device=device)
            call('ofx_set_range_words', ptr(t))
    return t


def reset_range_words(device):
    """Start of a sampling stage: the out-of-range count describes THIS stage only (the word is sticky, and a
    count left by an earlier training forward / test would fail the next sample_loop and switch the process to bf16x3)."""
    t = _RANGE.get((device.type, device.index if device.index is not None else torch.cuda.current_device()))
    if t is not None:
        t.zero_()


def range_error(device, result=None):
    """{'operands_beyond_fp16': n, 'result_non_finite': bool} if operands left the fp16 range on `device` since the words
    were last cleared -- or `result` (a tensor the suspect launches produced) is non-finite: an operand beyond the range
    poisons everything downstream with NaN -- else None (host sync)."""
    t = _RANGE.get((device.type, device.index if device.index is not None else torch.cuda.current_device()))
    n = int(t[0].item()) if t is not None else 0
    bad = result is not None and not bool(torch.isfinite(result).all())
    return {'operands_beyond_fp16': n, 'result_non_finite': bad} if (n or bad) else None


def raise_on_range_error(device, result=None):
    """Production check, once per sampling stage next to raise_on_sync_error.  With the fp16-pair contraction active,
    operands beyond +-65504 turn the result into NaN (they are not clamped): a non-finite `result`, or a non-zero
    out-of-range count, means this call's results are invalid.  Clears the words and raises OfxRangeError; with
    AUTO_RANGE_FALLBACK the process is switched to 'bf16x3' (fp32's exponent range, 16 significand bits) first, so
    that the caller's retry (pipeline.CascadeSampler.sample retries once by itself) succeeds.  In any other precision
    a non-finite result is the model's own (diverged weights / inputs): OfxError."""
    err = range_error(device, result)
    if err is None:
        return
    range_words(device).zero_()
    was = get_precision()
    if was not in ('fp16x3', 'fp16'):
        raise _lib.OfxError('non-finite result on %s in precision %s (%s)' % (device, was, err))
    if AUTO_RANGE_FALLBACK and was == 'fp16x3':
        set_precision('bf16x3')
    raise OfxRangeError('fp16x3 range guard on %s: %s -- operands beyond +-65504 do not fit the fp16 operand pairs; the '
                        'results of this call are invalid%s' % (device, err, '; the process now computes in bf16x3: call again'
                                                               if AUTO_RANGE_FALLBACK and was == 'fp16x3' else ''))


def workspace(device, nbytes=96 << 20):
    """Split-K / partial-tile scratch, one buffer per (device, stream): launches on one stream are ordered and reuse it,
    launches on different streams (side-stream experiment, graphs captured on their own stream) must not share it."""
    key = (device.type, device.index, _stream_id(device))
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        _no_capture('the split-K / hand-off workspace')
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


def scan_i32(x):
    """Exclusive scan; returns int32 tensor of n+1 entries (last = total)."""
    _chk(x, torch.int32)
    n = x.numel()
    out = torch.empty(n + 1, dtype=torch.int32, device=x.device)
    ws = torch.empty(_lib.lib().ofx_scan_ws_bytes(n), dtype=torch.uint8, device=x.device)
    call('ofx_scan_i32', ptr(x), ptr(out), n, ptr(ws), stream())
    return out


class PackedWeight:
    """Weights re-laid-out once for the MFMA B tile ([k/4][n][4], k zero-padded to 32)."""

    def __init__(self):
        self.t = None
        self.key = None
        self.K = self.N = self.Kp = 0

    def get(self, w, mode, cin=0, nt=0):
        """mode 'kn': w is [K, N]; 'nk': w is [N, K] (nn.Linear); 'graphconv': [7*(cin+nt'), N]."""
        # (the 16-bit planes behind the fp32 pack are bf16 or fp16 pairs, by the precision at pack time: part of the key)
        key = (w.data_ptr(), w._version, tuple(w.shape), mode, cin, nt, _lib.lib().ofx_get_precision() == 3)
        if key == self.key:
            return self
        _chk(w)
        w = w.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        L = _lib.lib()
        if mode == 'kn' or mode == 'graphconv':
            K, N = w.shape
            sk, sn = N, 1
        elif mode == 'nk':
            N, K = w.shape
            sk, sn = 1, K
        else:
            raise ValueError(mode)
        if mode == 'graphconv':
            Kp = L.ofx_graphconv_packed_k(cin, nt)
        else:
            cin = nt = 0
            Kp = L.ofx_packed_k(K)
        out = torch.empty(L.ofx_packed_floats(Kp, N), dtype=torch.float32, device=w.device)
        call('ofx_pack_weights', ptr(w), sk, sn, K, N, cin, nt, ptr(out), Kp, stream())
        self.t, self.key, self.K, self.N, self.Kp = out, key, K, N, Kp
        return self


def gemm(a, pw, bias=None, res=None, out=None, a_rows=None, out_rows=None, m=None, out_planes=0):
    """out[orow(i)] = a[arow(i)] @ W + bias + res[i] (fp32 MFMA).  out_planes (2 / 3): the rows are written as hi / lo
    pair planes (the operand format of the planes GraphConv) instead of fp32."""
    a, lda = _row_major(a)
    M = m if m is not None else (a_rows.numel() if a_rows is not None else a.shape[0])
    if a.shape[1] != pw.K:
        raise ValueError('K mismatch: %d vs %d' % (a.shape[1], pw.K))
    if out is None:
        out = torch.empty(M, pw.N, dtype=torch.float32, device=a.device)
    out2, ldc = _row_major(out)
    assert out2 is out, 'output must have contiguous rows'
    ldr = 0
    if res is not None:
        res, ldr = _row_major(res)
    _chk(bias)
    _chk(a_rows, torch.int32)
    _chk(out_rows, torch.int32)
    ws = workspace(a.device)
    _meta('dense_gemm', 2.0 * M * pw.K * pw.N, 4.0 * (M * pw.K + pw.K * pw.N + M * pw.N * (2 if res is not None else 1)),
          (M, pw.K, pw.N))
    if out_planes:
        args = ('ofx_gemm_f32_planes', ptr(a), lda, ptr(a_rows), M, pw.K, ptr(pw.t), pw.Kp, pw.N, ptr(bias),
                ptr(res), ldr, ptr(out), ldc, ptr(out_rows), ptr(ws), ws.numel(), out_planes, stream())
    else:
        args = ('ofx_gemm_f32', ptr(a), lda, ptr(a_rows), M, pw.K, ptr(pw.t), pw.Kp, pw.N, ptr(bias),
                ptr(res), ldr, ptr(out), ldc, ptr(out_rows), ptr(ws), ws.numel(), stream())
    if pw.N <= 64 or pw.K <= 64:
        with policy_scope('small_gemm'):
            call(*args)
    else:
        call(*args)
    return out


def gather_gemm(x, tab, ntap, pw, n_out, bias=None, res=None, out=None, out_rows=None, out_planes=0):
    """out[orow(r)] = concat_j x[tab[r, j]] @ W (+ bias + res[r]): the branch-free gather-GEMM with a caller-made table
    (ofx_gather_gemm_f32) -- Downsample on an x whose rows are not contiguous."""
    x, ldx = _row_major(x)
    cin = x.shape[1]
    assert pw.K == ntap * cin and cin % 32 == 0
    _chk(tab, torch.int32)
    if out is None:
        out = torch.empty(n_out, pw.N, dtype=torch.float32, device=x.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    ldr = 0
    if res is not None:
        res, ldr = _row_major(res)
    _chk(bias)
    _chk(out_rows, torch.int32)
    ws = workspace(x.device)
    _meta('dense_gemm', 2.0 * n_out * pw.K * pw.N, 4.0 * (n_out * pw.K + pw.K * pw.N + n_out * pw.N), (n_out, pw.K, pw.N))
    call('ofx_gather_gemm_f32', ptr(x), ldx, cin, ntap, x.shape[0], n_out, ptr(tab), ptr(zero_row(x.device)), ptr(pw.t), pw.Kp,
         pw.N, ptr(bias), ptr(res), ldr, ptr(out), ldc, ptr(out_rows), ptr(ws), ws.numel(), out_planes, stream())
    return out


# the unpool GEMM on the planes data path (ofx_gemm_planes).  Built in round 6 and measured at parity with the register-staged
# kernel it was to replace (tools/gemm_planes_probe.py: 94.0 vs 94.5 us at [4976, 512] x [512, 4096], 128 vs 118 us at
# [21344, 256] x [256, 2048], 390 vs 402 us at [71088, 256] x [256, 2048] incl. the 10-23 us gather + split of its input rows):
# off by default, kept for the A/B (OFX_GEMM_PLANES=1)
GEMM_PLANES = os.environ.get('OFX_GEMM_PLANES', '0') == '1'
_ROW_TAB = {}


class PackedGemmPlanes:
    """Dense weights [K, N] packed for ofx_gemm_planes (cached per parameter version and mode)."""

    def __init__(self):
        self.t = None
        self.key = None

    def get(self, w2d, mode):
        """w2d: a [K, N] VIEW of the parameter (any strides)."""
        key = (w2d.data_ptr(), w2d._version, tuple(w2d.shape), tuple(w2d.stride()), mode)
        if key != self.key:
            _chk(w2d)
            K, N = w2d.shape
            nb = _lib.lib().ofx_gemm_planes_packed_bytes(K, N, mode)
            assert nb > 0
            t = torch.empty(nb, dtype=torch.uint8, device=w2d.device)
            assert t.data_ptr() % 128 == 0
            call('ofx_pack_gemm_planes', ptr(w2d.detach()), w2d.stride(0), w2d.stride(1), K, N, mode, ptr(t), stream())
            self.t, self.key, self.K, self.N = t, key, K, N
        return self


def gemm_planes(a_planes, pgp, out, out_planes=0, bias=None):
    """out = a_planes @ W on the planes GraphConv's data path (ofx_gemm_planes; a_planes: pair planes [M, K], row m of the
    output from row m of the input).  Returns False when the shape does not qualify (nothing launched)."""
    mode = planes_of(a_planes)
    M, K = a_planes.shape
    assert planes_pairs(mode) and K == pgp.K and a_planes.stride(1) == 1
    out2, ldc = _row_major(out)
    assert out2 is out
    dev = a_planes.device
    key = (dev.index, M)
    tab = _ROW_TAB.get(key)
    if tab is None:
        if len(_ROW_TAB) > 64:
            _ROW_TAB.clear()
        _no_capture('the row table of a dense planes GEMM')
        tab = torch.zeros(M * 7 + 4, dtype=torch.int32, device=dev)
        tab[:M * 7].view(M, 7)[:, 0] = torch.arange(M, dtype=torch.int32, device=dev)
        _ROW_TAB[key] = tab
    ws = workspace(dev)
    sync = sync_words(dev)
    _meta('dense_gemm', 2.0 * M * K * pgp.N, 4.0 * (M * K + K * pgp.N + M * pgp.N), (M, K, pgp.N, 'planes'))
    prof = _lib.PROFILE
    lib = _lib.lib()
    args = (ptr(a_planes), a_planes.stride(0) * 4, M, M, K, ptr(tab), ptr(pgp.t), pgp.N, ptr(bias), ptr(out), ldc, out_planes,
            ptr(ws), ws.numel(), ptr(sync), sync.numel() * 4, mode, stream())
    if prof is not None:
        meta, _lib.META = _lib.META, None
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.ofx_gemm_planes(*args)
        e1.record()
        if rc == 0:
            prof.append(('ofx_gemm_planes', e0, e1, meta))
    else:
        _lib.META = None
        rc = lib.ofx_gemm_planes(*args)
    if rc < 0:
        raise _lib.OfxError('ofx_gemm_planes failed with status %d' % rc)
    return rc == 0


LINEAR_SMALL = True          # A/B switch: False sends the few-row linears through the MFMA GEMM again


def linear_small(a, weight, bias=None, res=None, act_in=None, act_out=None, out=None):
    """out = act_out(act_in(a) @ weight^T + bias + res) for a [M <= 16, K], weight [N, K] (nn.Linear layout, unpacked):
    ONE launch, exact fp32 (ofx_linear_small)."""
    a, lda = _row_major(a)
    w, ldw = _row_major(weight.detach())
    M, K = a.shape
    N = w.shape[0]
    assert w.shape[1] == K and M <= 16
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    out2, ldo = _row_major(out)
    assert out2 is out
    ldr = 0
    if res is not None:
        res, ldr = _row_major(res)
    _chk(bias)
    _meta('small_linear', 2.0 * M * K * N, 4.0 * (N * K + M * K + M * N), (M, K, N))
    call('ofx_linear_small', ptr(a), lda, M, K, ptr(w), ldw, N, ptr(bias.detach() if bias is not None else None), ptr(res),
         ldr, ACT[act_in], ACT[act_out], ptr(out), ldo, stream())
    return out


def learned_sinusoid(t, w):
    """[B, 2 * half + 1] = [t, sin(2 pi t w), cos(2 pi t w)] (LearnedSinusoidalPosEmb, modules.py:550-563)."""
    _chk(t), _chk(w)
    t, w = t.contiguous(), w.detach().contiguous()
    out = torch.empty(t.shape[0], 2 * w.shape[0] + 1, dtype=torch.float32, device=t.device)
    _meta('elementwise', 0, 4.0 * out.numel(), (out.numel(),))
    call('ofx_learned_sinusoid', ptr(t), ptr(w), t.shape[0], w.shape[0], ptr(out), stream())
    return out


_ZEROS = {}


def zero_row(device, n=4096):
    key = (device.type, device.index)
    if key not in _ZEROS:
        _ZEROS[key] = torch.zeros(n, dtype=torch.float32, device=device)
    return _ZEROS[key]


def graphconv(x, nbr, seg_ptr, col, pw, cin, type_frac=None, bias=None, emb=None, batch_id=None,
              res=None, out=None, ext=None, stats=None):
    """ext = (nbr_ext, multi_seg, n_multi) enables the branch-free fast path (cin % 32 == 0).
    stats (fp64 [B, Cout, 2], zeroed, needs batch_id): the epilogue also accumulates the GroupNorm
    statistics of the output."""
    """Fused dual-octree graph convolution (gather -> segment mean -> MFMA contraction)."""
    x, ldx = _row_major(x)
    N = x.shape[0]
    if x.shape[1] != cin:
        raise ValueError('cin mismatch')
    if out is None:
        out = torch.empty(N, pw.N, dtype=torch.float32, device=x.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    ldt = nt_pad = 0
    if type_frac is not None:
        _chk(type_frac)
        ldt = type_frac.stride(0)
        nt_pad = type_frac.shape[1]
    lde = ldr = 0
    if emb is not None:
        emb, lde = _row_major(emb)
    if emb is not None or stats is not None:
        _chk(batch_id, torch.int32)
    if res is not None:
        res, ldr = _row_major(res)
    _chk(bias)
    _chk(stats, torch.float64)
    prof = GRAPHCONV_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(x.device)
    nbr_ext = multi_seg = aux = None
    n_multi = 0
    if ext is not None and cin % 32 == 0:
        nbr_ext, multi_seg, n_multi = ext
        aux = torch.empty((n_multi + 1) * ldx, dtype=torch.float32, device=x.device)
    args = ('ofx_graphconv_fwd', ptr(x), ldx, cin, N, ptr(nbr), ptr(seg_ptr), ptr(col), ptr(nbr_ext), ptr(multi_seg),
            n_multi, ptr(aux), ptr(type_frac), ldt, nt_pad,
            ptr(pw.t), pw.Kp, pw.N, ptr(bias), ptr(emb), lde,
            ptr(batch_id) if (emb is not None or stats is not None) else None,
            ptr(res), ldr, ptr(out), ldc, ptr(stats), pw.N, ptr(ws), ws.numel(), stream())
    _meta('graphconv', 2.0 * N * pw.K * pw.N, 0, (N, cin, pw.N, 'reg'))
    if pw.N <= 64 or cin <= 64:
        with policy_scope('small_gemm'):
            call(*args)
    else:
        call(*args)
    if prof is not None:
        e1.record()
        E = col.numel()
        k_logical = pw.K                                   # 7 * (cin + nt)
        flops = 2.0 * N * k_logical * pw.N
        # fused-op algorithmic bytes: one feature row per edge + output + weights + 8 B/edge
        nbytes = 4.0 * (E * cin + N * pw.N + k_logical * pw.N) + 8.0 * E
        prof.append((e0, e1, flops, nbytes, pw.N, ('graph', N, cin, pw.N)))
    return out


NARROW_IN = os.environ.get('OFX_NARROW_IN', '1') == '1'      # A/B switches of the two gather-shaped GraphConvs (csrc/ofx_narrow.hip)
NARROW_OUT = os.environ.get('OFX_NARROW_OUT', '1') == '1'


def narrow_in_ok(cin, cout, nt):
    return NARROW_IN and cin <= 8 and cout in (64, 128) and 7 * (cin + nt) <= 96


# the table-driven, persistent, pipelined launch of the input convolution (ofx_graphconv_narrow_in_tab).  Measured (tools/
# narrow_in_probe.py, eager): depth 8 (3.25 M rows) 915 -> 588 us; depth 7 (0.71 M) 143 -> 138 us; depth 6 (0.22 M) 77 -> 81 us
# (kernel time 75 -> 65 us, but its two record pre-pass launches cost more than that on a tensor this short), so it takes
# the launches from NARROW_IN_TAB_MIN_ROWS rows up (OFX_NARROW_IN_TAB=0: never)
NARROW_IN_TAB = os.environ.get('OFX_NARROW_IN_TAB', '1') == '1'
NARROW_IN_TAB_MIN_ROWS = int(os.environ.get('OFX_NARROW_IN_TAB_MIN_ROWS', str(1 << 19)))


def graphconv_narrow_in(x, seg_ptr, col, weights, cin, nt, node_type=None, bias=None, batch_id=None, out=None, stats=None,
                        ext=None):
    """The U-Net's INPUT GraphConv (3 / 8 channels -> 64 / 128): gather + exact-fp32 FMA with the weights in registers
    (ofx_graphconv_narrow_in).  weights: the raw nn.Parameter [7 * (cin + nt), cout]; node_type: uint8 [N].
    ext = (nbr_ext, multi_seg, n_multi): the branch-free gather table -> the persistent, pipelined launch."""
    x, ldx = _row_major(x)
    w = weights.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    _chk(w)
    N, cout = x.shape[0], w.shape[1]
    assert x.shape[1] == cin and w.shape[0] == 7 * (cin + nt)
    if out is None:
        out = torch.empty(N, cout, dtype=torch.float32, device=x.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    if nt:
        _chk(node_type, torch.uint8)
    _chk(bias)
    _chk(stats, torch.float64)
    if stats is not None:
        _chk(batch_id, torch.int32)
    prof = GRAPHCONV_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    ws = workspace(x.device)
    E = col.numel()
    flops = 2.0 * N * w.shape[0] * cout
    nbytes = 4.0 * (E * cin + N * cout + w.numel()) + 8.0 * E
    _meta('graphconv_narrow', flops, nbytes, (N, cin, cout, 'narrow_in'))
    if ext is not None and NARROW_IN_TAB and N >= NARROW_IN_TAB_MIN_ROWS:
        nbr_ext, multi_seg, n_multi = ext
        aux = torch.empty((N + n_multi + 1) * (8 if cin <= 4 else 16), dtype=torch.float32, device=x.device)
        call('ofx_graphconv_narrow_in_tab', ptr(x), ldx, cin, N, ptr(seg_ptr), ptr(col), ptr(nbr_ext), ptr(multi_seg), n_multi,
             ptr(aux), ptr(node_type) if nt else None, nt, ptr(w), cout, ptr(bias),
             ptr(batch_id) if stats is not None else None, ptr(out), ldc, ptr(stats), cout, ptr(ws), ws.numel(), stream())
    else:
        call('ofx_graphconv_narrow_in', ptr(x), ldx, cin, N, ptr(seg_ptr), ptr(col), ptr(node_type) if nt else None, nt,
             ptr(w), cout, ptr(bias), ptr(batch_id) if stats is not None else None, ptr(out), ldc, ptr(stats), cout,
             ptr(ws), ws.numel(), stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, flops, nbytes, cout, ('graph', N, cin, cout)))
    return out


class PackedNarrowOut:
    """Wd [C, pw] of the project-then-aggregate output GraphConv (ofx_narrow_out_pack) + its MFMA pack."""

    def __init__(self):
        self.key = None
        self.pw = PackedWeight()

    def get(self, w, C, nt):
        key = (w.data_ptr(), w._version, tuple(w.shape), C, nt)
        if key != self.key:
            wd = w.detach()
            if not wd.is_contiguous():
                wd = wd.contiguous()
            _chk(wd)
            cout = wd.shape[1]
            self.width = 32 if 7 * cout <= 32 else 64
            self.wd = torch.empty(C, self.width, dtype=torch.float32, device=w.device)
            call('ofx_narrow_out_pack', ptr(wd), C, nt, cout, self.width, ptr(self.wd), stream())
            self.w = wd
            self.key = key
        return self

    def type_term(self, type_frac, nt, N, C, bias):
        """[N, cout] = bias + the node-type part of the convolution: constant per (graph depth, weights) -- the caller
        caches it on the doctree (ofx_narrow_out_type_term)."""
        cout = self.w.shape[1]
        tt = torch.empty(N, cout, dtype=torch.float32, device=self.w.device)
        _chk(bias)
        call('ofx_narrow_out_type_term', ptr(type_frac) if nt else None, type_frac.stride(0) if nt else 0, nt, N, ptr(self.w),
             C, cout, ptr(bias), ptr(tt), stream())
        return tt


def graphconv_narrow_out(x, seg_ptr, col, pno, C, type_term=None, out=None):
    """The U-Net's OUTPUT GraphConv (C channels -> 3 / 8) as project-then-aggregate: one dense GEMM P = x @ Wd (x read once,
    coalesced), then a gather of cout floats per edge (ofx_graphconv_narrow_out)."""
    N = x.shape[0]
    cout = pno.w.shape[1]
    prof = GRAPHCONV_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    P = gemm(x, pno.pw.get(pno.wd, 'kn'))
    if out is None:
        out = torch.empty(N, cout, dtype=torch.float32, device=x.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    _chk(type_term)
    assert type_term is None or (type_term.is_contiguous() and tuple(type_term.shape) == (N, cout))
    E = col.numel()
    flops = 2.0 * N * pno.w.shape[0] * cout
    # bytes of the algorithm that RUNS (project-then-aggregate), each array once: the aggregation reads P, the CSR
    # (7 N segment bounds + E columns) and the cached node-type / bias term and writes the output; the whole operator
    # (`prof` bracket: projection GEMM + aggregation) also reads x and writes P.  (Until round 5 this line carried the
    # reference operator's 4 E C bytes, which this path never moves: its "fraction of the HBM roof" read 2.0.)
    pwid = P.shape[1]
    agg_bytes = 4.0 * (N * pwid + E + 7 * N + 2 * N * cout)
    nbytes = agg_bytes + 4.0 * (N * C + N * pwid + pno.w.numel())
    _meta('graphconv_narrow', flops, agg_bytes, (N, C, cout, 'narrow_out'))
    call('ofx_graphconv_narrow_out', ptr(P), P.stride(0), cout, N, ptr(seg_ptr), ptr(col), ptr(type_term), ptr(out), ldc,
         stream())
    if prof is not None:
        e1.record()
        prof.append((e0, e1, flops, nbytes, cout, ('graph', N, C, cout)))
    return out


def gemm_tn(p, q):
    """p^T @ q for row-major p [rows, K], q [rows, N] (dense weight gradients: dW = x^T dy)."""
    _chk(p), _chk(q)
    assert p.shape[0] == q.shape[0]
    K0, N0 = p.shape[1], q.shape[1]

    def pad4(t):                       # odd widths (e.g. the 17-wide sinusoid embedding): zero-pad to 4 columns
        c = t.shape[1]
        if c % 4 == 0 and t.stride(0) % 4 == 0 and t.stride(1) == 1 and t.data_ptr() % 16 == 0:
            return t
        out = torch.zeros(t.shape[0], (c + 3) // 4 * 4, dtype=torch.float32, device=t.device)
        out[:, :c] = t
        return out
    p, q = pad4(p), pad4(q)
    if (K0, N0) != (p.shape[1], q.shape[1]):
        return gemm_tn(p, q)[:K0, :N0].contiguous()
    p, ldp = _row_major(p)
    q, ldq = _row_major(q)
    K, N = p.shape[1], q.shape[1]
    out = torch.empty(K, N, dtype=torch.float32, device=p.device)
    ws = workspace(p.device)
    call('ofx_gemm_tn_f32', ptr(p), ldp, ptr(q), ldq, p.shape[0], K, N, ptr(out), ptr(ws), ws.numel(), stream())
    return out


def linear_backward(x, dy, weight, need_dx=True):
    """y = x @ weight^T + bias (nn.Linear / Conv1x1, weight [out, in]): returns (dx, dW [out, in], dbias [out])."""
    dW = gemm_tn(dy, x)
    dx = gemm(dy, PackedWeight().get(weight, 'kn')) if need_dx else None        # dy [n, out] @ W [out, in]
    return dx, dW, dy.sum(0)


def graphconv_backward(x, dy, doctree, d, weights, n_node_type, need_dx=True, need_dw=True):
    """Gradients of y = GraphConv(x) (modules.py:194-220) w.r.t. x and the weights, given dy = dL/dy.
    Returns (dx [N, Cin] or None, dW [7*(Cin+nt'), Cout] or None) -- training path, SURVEY 8f-4."""
    x, ldx = _row_major(x)
    dy, ldy = _row_major(dy)
    _chk(x), _chk(dy), _chk(weights)
    N, cin = x.shape
    cout = dy.shape[1]
    nt = n_node_type if n_node_type > 1 else 0
    assert weights.shape == (7 * (cin + nt), cout)
    seg_ptr, col, Ng, E = doctree.csr(d)
    assert N == Ng and dy.shape[0] == N
    ws = workspace(x.device)
    dx = dW = None
    if need_dx:
        rv = doctree.rev(d)
        wt = weights.detach().view(7, cin + nt, cout)[:, :cin, :].permute(0, 2, 1).reshape(7 * cout, cin).contiguous()
        pwt = PackedWeight().get(wt, 'graphconv', cout, 0)
        dx = torch.empty(N, cin, dtype=torch.float32, device=x.device)
        fast = cout % 32 == 0 and ldy % 4 == 0
        aux = torch.empty((rv['V'] + 1) * ldy, dtype=torch.float32, device=x.device) if fast else None
        call('ofx_graphconv_bwd_data', ptr(dy), ldy, cout, N, ptr(rv['nbr']), ptr(rv['rev_ptr']), ptr(rv['rev_row']),
             ptr(rv['rev_w']), ptr(rv['nbr_ext']) if fast else None, ptr(rv['multi_seg']) if fast else None,
             rv['V'] if fast else 0, ptr(aux), ptr(pwt.t), pwt.Kp, cin, ptr(dx), cin, ptr(ws), ws.numel(), stream())
    if need_dw:
        L = _lib.lib()
        Kp = L.ofx_graphconv_packed_k(cin, nt)
        Kf = Kp - (((7 * nt + 31) // 32) * 32 if nt else 0)
        tf = doctree.type_frac(d, nt) if nt else None
        fast = cin % 32 == 0 and ldx % 4 == 0
        nbr_ext, multi_seg, V = doctree.ext(d)
        aux = torch.empty((V + 1) * ldx, dtype=torch.float32, device=x.device) if fast else None
        dyw, ldw, cw = dy, ldy, cout
        if cout % 4 or ldy % 4 or dy.data_ptr() % 16:      # narrow outputs (the 3-channel output conv): pad to 4 columns
            cw = (cout + 3) // 4 * 4
            dyw = torch.zeros(N, cw, dtype=torch.float32, device=x.device)
            dyw[:, :cout] = dy
            ldw = cw
        dwp = torch.empty(Kp, cw, dtype=torch.float32, device=x.device)
        call('ofx_graphconv_bwd_weight', ptr(x), ldx, cin, N, ptr(doctree.nbr(d)), ptr(seg_ptr), ptr(col),
             ptr(nbr_ext) if fast else None, ptr(multi_seg) if fast else None, V if fast else 0, ptr(aux),
             ptr(tf), tf.stride(0) if tf is not None else 0, tf.shape[1] if tf is not None else 0,
             ptr(dyw), ldw, cw, ptr(dwp), Kp, ptr(ws), ws.numel(), stream())
        dwp = dwp[:, :cout]
        # packed k order -> the reference's row order dir*(cin+nt) + [channels | types]
        dev = x.device
        dirs = torch.arange(7, device=dev).view(7, 1)
        idx = [dirs * cin + torch.arange(cin, device=dev).view(1, cin)]
        if nt:
            idx.append(Kf + dirs * nt + torch.arange(nt, device=dev).view(1, nt))
        dW = dwp.index_select(0, torch.cat(idx, 1).reshape(-1))
    return dx, dW


class PackedConv3d:
    """nn.Conv3d weight [cout, cin, 3, 3, 3] packed for the 27-tap gather-GEMM."""

    def __init__(self):
        self.t = None
        self.key = None
        self.cin = self.N = 0

    def get(self, w):
        key = (w.data_ptr(), w._version, tuple(w.shape), _lib.lib().ofx_get_precision() == 3)
        if key == self.key:
            return self
        _chk(w)
        cout, cin = w.shape[:2]
        assert tuple(w.shape[2:]) == (3, 3, 3)
        w = w.detach().contiguous()
        Kp = _lib.lib().ofx_conv3d_packed_k(cin)
        out = torch.empty(_lib.lib().ofx_packed_floats(Kp, cout), dtype=torch.float32, device=w.device)
        call('ofx_pack_conv3d', ptr(w), cin, cout, ptr(out), stream())
        self.t, self.key, self.cin, self.N = out, key, cin, cout
        return self


def gridconv(x, tables, n_out, pw, bias=None, emb=None, batch_id=None, res=None, out=None):
    """3x3x3 convolution on a dense octree layer in node-row layout (27-tap gather-GEMM).
    tables(fast) -> the neighbour table padded with n_in (fast) or -1 (generic)."""
    x, ldx = _row_major(x)
    if x.shape[1] != pw.cin:
        raise ValueError('cin mismatch')
    if out is None:
        out = torch.empty(n_out, pw.N, dtype=torch.float32, device=x.device)
    out2, ldc = _row_major(out)
    assert out2 is out
    lde = ldr = 0
    if emb is not None:
        emb, lde = _row_major(emb)
        _chk(batch_id, torch.int32)
    if res is not None:
        res, ldr = _row_major(res)
    _chk(bias)
    ws = workspace(x.device)
    fast = pw.cin % 32 == 0 and ldx % 4 == 0
    prof = GRAPHCONV_PROFILE
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    _meta('gridconv_27tap', 2.0 * n_out * 27 * pw.cin * pw.N,
          4.0 * (x.shape[0] * pw.cin + n_out * pw.N * (2 if res is not None else 1) + 27.0 * pw.cin * pw.N) + 4.0 * 27 * n_out,
          (n_out, pw.cin, pw.N))
    call('ofx_gridconv_fwd', ptr(x), ldx, pw.cin, x.shape[0], n_out, None if fast else ptr(tables(False)),
         ptr(tables(True)) if fast else None, ptr(zero_row(x.device)), ptr(pw.t), pw.N, ptr(bias), ptr(emb), lde,
         ptr(batch_id) if emb is not None else None, ptr(res), ldr, ptr(out), ldc, ptr(ws), ws.numel(), stream())
    if prof is not None:
        e1.record()
        # same kernel as the GraphConv (27 taps): algorithmic = one source row per tap + output + weights + indices
        flops = 2.0 * n_out * 27 * pw.cin * pw.N
        nbytes = 4.0 * (27.0 * n_out * pw.cin + n_out * pw.N + 27.0 * pw.cin * pw.N) + 4.0 * 27 * n_out
        prof.append((e0, e1, flops, nbytes, pw.N, ('grid', n_out, pw.cin, pw.N)))
    return out


def grid_conv_table(mode, depth_out, batch_size, device, pad=-1):
    n = (8 ** depth_out) * batch_size
    t = torch.empty(n, 27, dtype=torch.int32, device=device)
    call('ofx_grid_conv_table', mode, depth_out, batch_size, pad, ptr(t), stream())
    return t


def attention(qkv, batch_size, T, heads, out=None):
    qkv, ldq = _row_major(qkv)
    C = qkv.shape[1] // 3
    ch = C // heads
    if out is None:
        out = torch.empty(qkv.shape[0], C, dtype=torch.float32, device=qkv.device)
    out2, ldo = _row_major(out)
    assert out2 is out
    _meta('attention', 4.0 * batch_size * heads * T * T * ch, 4.0 * qkv.shape[0] * 4 * C, (batch_size, T, heads, ch))
    call('ofx_attention', ptr(qkv), ldq, batch_size, T, heads, ch, ptr(out), ldo, stream())
    return out


def attention_backward(qkv, dout, batch_size, T, heads):
    """dqkv of out = attention(qkv) (QKVAttention) given dout -- training path."""
    qkv, ldq = _row_major(qkv)
    dout, ldo = _row_major(dout)
    _chk(qkv), _chk(dout)
    C = qkv.shape[1] // 3
    ch = C // heads
    dqkv = torch.empty(qkv.shape[0], 3 * C, dtype=torch.float32, device=qkv.device)
    rowstat = torch.empty(batch_size * heads * T * 3, dtype=torch.float32, device=qkv.device)
    call('ofx_attention_bwd', ptr(qkv), ldq, ptr(dout), ldo, batch_size, T, heads, ch, ptr(rowstat), ptr(dqkv), 3 * C,
         stream())
    return dqkv


def gather_mean(x, seg_ptr, col):
    """col_data [N, 7, C]: the reference's scatter_mean(x[col], row*7+dir)."""
    x, ldx = _row_major(x)
    N, C = x.shape
    out = torch.empty(N, 7, C, dtype=torch.float32, device=x.device)
    call('ofx_gather_mean', ptr(x), ldx, C, N, ptr(seg_ptr), ptr(col), ptr(out), stream())
    return out


AUX_ATTR = '_ofx_aux'
GN_FINALIZE_LAUNCH = False
# dense grids with at least this many rows per batch element take the statistics + apply pair instead of the one-launch
# kernel (whose grid is batch x channel slices: 32 blocks for 16^3 x 64 at batch 8) -- A/B: OFX_GN_ROWS_TWO_KERNELS
# (round 4, same box, same run: lr stage 1.052 -> 1.015 ms per step, hr 8.80 -> 8.74 with 2048 = the 16^3 level)
GN_ROWS_TWO_KERNELS = int(os.environ.get('OFX_GN_ROWS_TWO_KERNELS', '2048'))
GN_FUSE_AUX_FINALIZE = os.environ.get('OFX_GN_FUSE_AUX_FINALIZE', '0') == '1'
# sibling-octet launch (ofx_gn_apply_planes_oct): derive mean / rstd per block inside the launch (no ofx_gn_finalize launch)
GN_OCT_FINALIZE = os.environ.get('OFX_GN_OCT_FINALIZE', '1') == '1'
# ... for tensors up to this many elements.  Every block repeats the finalize (64 threads x 2 cpg fp64 loads + a barrier before
# its first row can be used): measured against the 4.7 us ofx_gn_finalize launch it wins on small launches (depth 5, C = 128:
# 21.0 vs 28.2 us; depth 4, C = 256: 18.6 vs 27.2 us eager) and loses on long or wide ones (depth 6, C = 384: 181 vs 159 us;
# depth 5, C = 768: 114 vs 98 us -- 24 channels per group, 8-row blocks), tools/gn_probe_oct.py.  The one-shape step (B = 1)
# is almost entirely below the bar.
GN_OCT_FINALIZE_MAX_ELEMS = int(os.environ.get('OFX_GN_OCT_FINALIZE_MAX_ELEMS', str(9 << 20)))
# who writes the aux rows of a GroupNorm launch: 'oct' (default, round 6) = the thread that holds the sibling octet of their
# sources in registers (ofx_gn_apply_planes_oct); 'block' (round 4) = the 64-row block that holds their sources, from its own
# output; '' / '0' = separate aux blocks (round 3).  A/B: OFX_AUX_PLAN=oct|block|0
AUX_PLAN = {'1': 'oct', '0': ''}.get(os.environ.get('OFX_AUX_PLAN', 'oct'), os.environ.get('OFX_AUX_PLAN', 'oct'))


def group_norm(x, batch_id, count, batch_size, weight, bias, groups, eps=1e-5, act=None, out=None,
               count_eps=None, stats=None, rows_per_batch=None, planes=0, aux_graph=None):
    """DualOctreeGroupNorm (+ optional fused activation).  count_eps=0 gives torch.nn.GroupNorm.
    stats: fp64 [B, C, 2] sums already produced by the epilogue of the kernel that wrote x.
    rows_per_batch: every batch element owns that many CONTIGUOUS rows (dense grids) -> one fused launch."""
    if count_eps is None:
        count_eps = eps
    x, ldx = _row_major(x)
    n, C = x.shape
    dev = x.device
    range_words(dev)
    if (rows_per_batch is not None and stats is None and not planes and rows_per_batch * (C // groups) <= (1 << 16)
            and rows_per_batch < GN_ROWS_TWO_KERNELS):
        assert n == rows_per_batch * batch_size
        if out is None:
            out = torch.empty(n, C, dtype=torch.float32, device=dev)
        out2, ldo = _row_major(out)
        assert out2 is out
        _meta('gn_fused_rows', 0, 8.0 * n * C, (n, C))
        call('ofx_gn_fused_rows', ptr(x), ldx, rows_per_batch, batch_size, C, groups, eps, count_eps,
             ptr(weight.detach().reshape(-1)), ptr(bias.detach().reshape(-1)), ACT[act], ptr(out), ldo, stream())
        return out
    if stats is not None:
        _chk(stats, torch.float64)
        assert stats.numel() == batch_size * C * 2
        sums = stats
    else:
        _meta('gn_stats', 0, 4.0 * n * C, (n, C))
        if stats_pool_has_room(batch_size * C * 2, dev):
            sums = stats_zeros(batch_size * C * 2, dev)          # pre-zeroed pool slice: no memset node in the step
            call('ofx_gn_stats_acc', ptr(x), ldx, n, C, ptr(batch_id), batch_size, ptr(sums), stream())
        else:
            sums = torch.empty(batch_size * C * 2, dtype=torch.float64, device=dev)
            call('ofx_gn_stats', ptr(x), ldx, n, C, ptr(batch_id), batch_size, ptr(sums), stream())
    # mean / rstd are derived from the sums inside the apply launch (ofx.h: no ofx_gn_finalize launch) -- except when
    # the launch also writes the consuming GraphConv's aux rows: an aux block's work is a chain of dependent loads
    # (segment -> edge range -> column -> row) and deriving the statistics first makes the chain longer.  Measured twice:
    # round 3, every aux thread deriving its own statistics: 75 us vs 61 us at depth 6, C = 128; round 4, once per aux
    # block from its first / last segment's batch elements (the kernel still does that when no mean / rstd is passed):
    # gn_apply 1.15 -> 1.47 ms per hr step, against 0.115 ms for the 18 finalize launches it removes.  So that case keeps
    # the 4 us finalize launch.  GN_FINALIZE_LAUNCH = True forces it everywhere, GN_FUSE_AUX_FINALIZE = True removes it (A/B).
    mean = rstd = None
    oct_fin = (GN_OCT_FINALIZE and planes and aux_graph is not None and AUX_PLAN == 'oct' and len(aux_graph) > 4
               and aux_graph[4] is not None and len(aux_graph[4]) == 5 and n * C <= GN_OCT_FINALIZE_MAX_ELEMS)
    if GN_FINALIZE_LAUNCH or (planes and aux_graph is not None and not GN_FUSE_AUX_FINALIZE and not oct_fin):
        mean = torch.empty(batch_size * C, dtype=torch.float32, device=dev)
        rstd = torch.empty(batch_size * C, dtype=torch.float32, device=dev)
        _meta('gn_finalize', 0, 24.0 * batch_size * C, (batch_size, C))
        call('ofx_gn_finalize', ptr(sums), ptr(count), batch_size, C, groups, eps, count_eps, ptr(mean), ptr(rstd),
             stream())
    w = weight.detach().reshape(-1)
    b = bias.detach().reshape(-1)
    if planes:
        # the consumer is the planes GraphConv: write its operand format directly.  aux_graph = (seg_ptr, col,
        # multi_seg, n_multi) of the consumer's graph: the same launch then also writes its aux rows (zero row +
        # multi-neighbour means), which needs a destination other than x.
        if aux_graph is not None and out is not None and out.data_ptr() == x.data_ptr():
            out = None
        out, ldo = _planes_out(n, C, planes, dev, out if planes_pairs(planes) else None)
        aux = seg_ptr = col = multi_seg = plan = None
        n_multi = n_left = 0
        if aux_graph is not None:
            seg_ptr, col, multi_seg, n_multi = aux_graph[:4]
            oct_plan = None
            if len(aux_graph) > 4 and aux_graph[4] is not None and AUX_PLAN:
                if len(aux_graph[4]) == 5:
                    oct_plan = aux_graph[4]          # dual_octree.DualOctree.oct_plan
                else:
                    plan, n_left = aux_graph[4]      # (int32 plan, leftover count): dual_octree.DualOctree.aux_plan
            aux = torch.empty((n_multi + 1) * ldo, dtype=torch.uint8, device=dev)
        _meta('gn_apply', 0, 8.0 * n * C + (n_multi + 1) * float(ldo), (n, C, 'planes+aux' if aux is not None else 'planes'))
        if aux is not None and oct_plan is not None and (mean is not None or oct_fin):
            op, shift, n_own, n_left, (o_ptr, o_ent, o_head, o_src) = oct_plan
            base = op.data_ptr()
            call('ofx_gn_apply_planes_oct', ptr(x), ldx, n, C, ptr(batch_id), ptr(mean), ptr(rstd), ptr(sums), ptr(count),
                 groups, eps, count_eps, ptr(w), ptr(b), ACT[act], planes, ptr(out), ldo, n_multi, ptr(aux), base + 4 * o_ptr,
                 base + 4 * o_ent, n_own, shift, base + 4 * o_head, base + 4 * o_src, n_left, stream())
            setattr(out, PLANES_ATTR, planes)
            setattr(out, AUX_ATTR, aux)
            return out
        call('ofx_gn_apply_planes', ptr(x), ldx, n, C, ptr(batch_id), ptr(mean), ptr(rstd), ptr(sums), ptr(count), groups, eps,
             count_eps, ptr(w), ptr(b), ACT[act], planes, ptr(out), ldo, ptr(seg_ptr), ptr(col), ptr(multi_seg), n_multi,
             ptr(aux), ptr(plan), n_left, stream())
        setattr(out, PLANES_ATTR, planes)
        if aux is not None:
            setattr(out, AUX_ATTR, aux)
        return out
    if out is None:
        out = torch.empty(n, C, dtype=torch.float32, device=dev)
    out2, ldo = _row_major(out)
    assert out2 is out
    _meta('gn_apply', 0, 8.0 * n * C, (n, C, 'fp32'))
    call('ofx_gn_apply', ptr(x), ldx, n, C, ptr(batch_id), ptr(mean), ptr(rstd), ptr(sums), ptr(count), groups, eps, count_eps,
         ptr(w), ptr(b), ACT[act], ptr(out), ldo, stream())
    return out


def group_norm_backward(x, dy, batch_id, count, batch_size, weight, bias, groups, eps=1e-5, act=None, count_eps=None):
    """(dx, dgamma, dbeta) of y = act(DualOctreeGroupNorm(x)) given dy -- training path."""
    if count_eps is None:
        count_eps = eps
    x, ldx = _row_major(x)
    dy, ldy = _row_major(dy)
    _chk(x), _chk(dy)
    n, C = x.shape
    dev = x.device
    mean = torch.empty(batch_size * C, dtype=torch.float32, device=dev)
    rstd = torch.empty(batch_size * C, dtype=torch.float32, device=dev)
    sums = torch.empty(batch_size * C * 2, dtype=torch.float64, device=dev)
    call('ofx_gn_stats', ptr(x), ldx, n, C, ptr(batch_id), batch_size, ptr(sums), stream())
    call('ofx_gn_finalize', ptr(sums), ptr(count), batch_size, C, groups, eps, count_eps, ptr(mean), ptr(rstd),
         stream())
    w = weight.detach().reshape(-1).contiguous()
    b = bias.detach().reshape(-1).contiguous()
    coef = torch.empty(batch_size * C * 3, dtype=torch.float32, device=dev)
    dx = torch.empty(n, C, dtype=torch.float32, device=dev)
    dgamma = torch.empty(C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(C, dtype=torch.float32, device=dev)
    call('ofx_gn_backward', ptr(x), ldx, ptr(dy), ldy, n, C, ptr(batch_id), batch_size, ptr(count), groups, count_eps,
         ptr(mean), ptr(rstd), ptr(w), ptr(b), ACT[act], ptr(sums), ptr(coef), ptr(dx), C, ptr(dgamma), ptr(dbeta),
         stream())
    return dx, dgamma, dbeta


STATS_ATTR = '_ofx_gn_stats'


class _StatsPool:
    """Zero-initialised fp64 slices for the fused GroupNorm statistics: one fill per network forward instead
    of one per GraphConv (each tiny fill costs a full launch slot on the stream).  One pool per (device, STREAM), like
    the other scratch: forwards on different streams (the lanes of sampler.sample_loop, graphs captured on their own
    stream) run concurrently and must not zero or accumulate into each other's slices."""
    SIZE = 1 << 19                      # doubles (4 MB)

    def __init__(self):
        self.buf = None
        self.cur = 0
        self.depth = 0
        self.gen = 0


_stats_pools = {}


def _stats_pool_of(device):
    key = (device.type, device.index, _stream_id(device))
    p = _stats_pools.get(key)
    if p is None:
        p = _stats_pools[key] = _StatsPool()
    return p


@contextlib.contextmanager
def stats_scope(device):
    """Buffers handed out by stats_zeros() inside the (outermost) scope come from one pre-zeroed pool; they are
    valid until the next outermost scope on the same stream begins (get_stats() drops stale ones)."""
    p = _stats_pool_of(device)
    if p.depth == 0:
        if p.buf is None:
            _no_capture('the GroupNorm statistics pool')
            p.buf = torch.empty(p.SIZE, dtype=torch.float64, device=device)
        p.buf.zero_()
        p.cur = 0
        p.gen += 1
    p.depth += 1
    try:
        yield
    finally:
        p.depth -= 1


def stats_pool_has_room(n, device):
    p = _stats_pool_of(device)
    return p.depth > 0 and p.cur + n <= p.SIZE


def stats_zeros(n, device):
    p = _stats_pool_of(device)
    if p.depth > 0 and p.cur + n <= p.SIZE:
        st = p.buf[p.cur:p.cur + n]
        p.cur += (n + 1) & ~1           # keep 16-B alignment
        st._ofx_gen = p.gen
        st._ofx_pool = p
        return st
    return torch.zeros(n, dtype=torch.float64, device=device)


def get_stats(t):
    """GroupNorm sums attached to a tensor by the kernel epilogue that produced it (or None)."""
    st = getattr(t, STATS_ATTR, None)
    pool = getattr(st, '_ofx_pool', None)
    if pool is not None and st._ofx_gen != pool.gen:
        return None                     # pooled slice from an earlier forward: recycled since
    return st


def cat_channels(a, b, buf=None):
    """torch.cat([a, b], dim=1) that also concatenates attached GroupNorm statistics.  If `buf` is given,
    a and b are already the left / right column slices of it (zero-copy) and only the statistics are merged."""
    out = buf if buf is not None else torch.cat([a, b], dim=1)
    sa, sb = get_stats(a), get_stats(b)
    if sa is not None and sb is not None:
        B = sa.numel() // (a.shape[1] * 2)
        st = torch.cat([sa.view(B, a.shape[1], 2), sb.view(B, b.shape[1], 2)], dim=1).contiguous().view(-1)
        setattr(out, STATS_ATTR, st)
    return out


def rows_copy(src, dst, n, smap=None, dmap=None, C=None, planes=0):
    """dst[dmap(i)] = src[smap(i)]; planes (2 / 3): the destination rows are written as hi / lo pair planes."""
    src, lds = _row_major(src)
    dst2, ldd = _row_major(dst)
    assert dst2 is dst
    C = C if C is not None else src.shape[1]
    _meta('rows_copy', 0, 8.0 * n * C, (n, C))
    if planes:
        call('ofx_rows_copy_planes', ptr(src), lds, ptr(smap), ptr(dst), ldd, ptr(dmap), n, C, planes, stream())
    else:
        call('ofx_rows_copy', ptr(src), lds, ptr(smap), ptr(dst), ldd, ptr(dmap), n, C, stream())
    return dst


def act(x, kind, out=None):
    _chk(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty_like(x)
    _meta('elementwise', 0, 8.0 * x.numel(), (x.numel(),))
    call('ofx_act', ptr(x), ptr(out), x.numel(), ACT[kind], stream())
    return out


def timestep_embedding(t, dim, max_period=10000.0):
    _chk(t)
    t = t.contiguous()
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    call('ofx_timestep_embedding', ptr(t), t.shape[0], dim, float(max_period), ptr(out), stream())
    return out


def octree2voxel_cf(data, batch_size, depth):
    data, ld = _row_major(data)
    C = data.shape[1]
    S = 1 << depth
    vox = torch.empty(batch_size, C, S, S, S, dtype=torch.float32, device=data.device)
    call('ofx_octree2voxel_cf', ptr(data), ld, C, batch_size, depth, ptr(vox), stream())
    return vox


def voxel2octree_cf(vox, depth, out=None):
    _chk(vox)
    vox = vox.contiguous()
    B, C = vox.shape[:2]
    if out is None:
        out = torch.empty(B * (8 ** depth), C, dtype=torch.float32, device=vox.device)
    out2, ld = _row_major(out)
    assert out2 is out
    call('ofx_voxel2octree_cf', ptr(vox), C, B, depth, ptr(out), ld, stream())
    return out


def ddim_eps_update(x, eps, coef, x0_out=None):
    _chk(x), _chk(eps), _chk(coef), _chk(x0_out)
    assert x.is_contiguous() and eps.is_contiguous()
    _meta('elementwise', 0, 12.0 * x.numel(), (x.numel(),))
    call('ofx_ddim_eps_update', ptr(x), ptr(eps), ptr(coef), ptr(x0_out), x.numel(), stream())
    return x


def ddim_x0_update(x, x0, noise, coef):
    _chk(x), _chk(x0), _chk(coef)
    assert x.is_contiguous() and x0.is_contiguous()
    _meta('elementwise', 0, 16.0 * x.numel(), (x.numel(),))
    call('ofx_ddim_x0_update', ptr(x), ptr(x0), ptr(noise), ptr(coef), x.numel(), stream())
    return x
