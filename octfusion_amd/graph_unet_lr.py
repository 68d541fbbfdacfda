"""Dense 16^3 voxel U-Net -- stage "lr", also nested as the middle of "hr".

Mirror of reference models/networks/diffusion_networks/graph_unet_lr.py
(``UNet3DModel``: ctor :65-173, forward :184-230, forward_as_middle :175-182)
and its dense blocks (modules.py:26-95, 474-563): same constructor keywords,
signatures and state_dict keys (time_pos_emb.weights, time_emb.*, input_emb.*,
downs/ups.*, mid_*, end.*, out.* with nn.Conv3d / nn.GroupNorm / nn.Conv1d
parameter shapes).

MI355X design: the whole net runs in the octree's node-row layout.  A full
octree layer of depth d (16^3, 8^3, 4^3 ...) is rows ``b*8^d + morton(x,y,z)``
with channels contiguous, exactly what the sparse "hr" net hands over, so the
reference's octree2voxel / permute / gather-back (graph_unet_lr.py:176-181)
disappear.  Every 3x3x3 convolution -- stride 1, the stride-2 ConvDownsample
and the nearest-upsample+conv ConvUpsample (no upsampled tensor is ever
written) -- is the fused gather-GEMM of libofx with a 27-tap neighbour table
(built once per (batch, depth), independent of the octree); the time-embedding
add and the residual are fused into the conv epilogues; GroupNorm+SiLU is the
two-kernel libofx norm; attention is a libofx kernel.  No MIOpen / ATen compute.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .modules import _Linear


class _GridCache:
    """Per (batch, device): row-layout tables of the dense layers (octree independent)."""

    _cache = {}

    @classmethod
    def get(cls, batch_size, device):
        key = (batch_size, device.type, device.index)
        c = cls._cache.get(key)
        if c is None:
            c = cls(batch_size, device)
            cls._cache[key] = c
        return c

    def __init__(self, batch_size, device):
        self.B = batch_size
        self.device = device
        self._tab = {}
        self._bid = {}
        self._cnt = {}

    def table(self, mode, depth_out, fast):
        """27-tap neighbour table; out-of-grid taps = n_in (zero row; branch-free kernel) or -1 (generic)."""
        k = (mode, depth_out, fast)
        if k not in self._tab:
            d_in = depth_out + (0, 1, -1)[mode]
            pad = self.B * 8 ** d_in if fast else -1
            self._tab[k] = ops.grid_conv_table(mode, depth_out, self.B, self.device, pad)
        return self._tab[k]

    def rev(self, mode, depth_out):
        """Reverse tap tables of (mode, depth_out) for the backward pass of the 27-tap conv (ofx.h)."""
        k = ('rev', mode, depth_out)
        if k not in self._tab:
            from ._lib import call, ptr, stream, lib
            d_in = depth_out + (0, 1, -1)[mode]
            n_in, n_out = self.B * 8 ** d_in, self.B * 8 ** depth_out
            tab = self.table(mode, depth_out, False)
            dev = self.device
            nseg = n_in * 27
            cnt = torch.empty(nseg, dtype=torch.int32, device=dev)
            call('ofx_table_reverse_count', ptr(tab), n_out, 27, n_in, ptr(cnt), stream())
            rev_ptr = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
            ws = torch.empty(lib().ofx_scan_ws_bytes(nseg), dtype=torch.uint8, device=dev)
            call('ofx_scan_i32', ptr(cnt), ptr(rev_ptr), nseg, ptr(ws), stream())
            E = int(rev_ptr[-1].item())
            rev_row = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
            rev_w = torch.empty(max(E, 1), dtype=torch.float32, device=dev)
            call('ofx_table_reverse_fill', ptr(tab), n_out, 27, n_in, ptr(rev_ptr), ptr(cnt), ptr(rev_row), ptr(rev_w),
                 stream())
            nbr = torch.empty(nseg, dtype=torch.int32, device=dev)
            call('ofx_seg_primary_w', ptr(rev_ptr), ptr(rev_row), ptr(rev_w), nseg, ptr(nbr), stream())
            call('ofx_seg_multi_flag_w', ptr(rev_ptr), ptr(rev_w), nseg, ptr(cnt), stream())
            rank = torch.empty(nseg + 1, dtype=torch.int32, device=dev)
            call('ofx_scan_i32', ptr(cnt), ptr(rank), nseg, ptr(ws), stream())
            V = int(rank[-1].item())
            nbr_ext = torch.empty(nseg, dtype=torch.int32, device=dev)
            multi_seg = torch.empty(max(V, 1), dtype=torch.int32, device=dev)
            call('ofx_seg_primary_ext_w', ptr(rev_ptr), ptr(rev_row), ptr(rev_w), nseg, n_out, ptr(rank), ptr(nbr_ext),
                 ptr(multi_seg), stream())
            self._tab[k] = dict(rev_ptr=rev_ptr, rev_row=rev_row, rev_w=rev_w, nbr=nbr, nbr_ext=nbr_ext,
                                multi_seg=multi_seg, V=V, n_in=n_in, n_out=n_out)
        return self._tab[k]

    def batch_id(self, depth):
        if depth not in self._bid:
            per = 8 ** depth
            self._bid[depth] = (torch.arange(self.B * per, device=self.device) // per).to(torch.int32)
            self._cnt[depth] = torch.full((self.B,), float(per), dtype=torch.float32, device=self.device)
        return self._bid[depth]

    def count(self, depth):
        self.batch_id(depth)
        return self._cnt[depth]


class GridState:
    """What a dense block needs to know about its input: batch, depth, cached tables."""

    def __init__(self, batch_size, depth, device):
        self.B, self.depth = batch_size, depth
        self.cache = _GridCache.get(batch_size, device)

    def at(self, depth):
        g = GridState.__new__(GridState)
        g.B, g.depth, g.cache = self.B, depth, self.cache
        return g


def _dense_io(x):
    """Reference-signature entry of the dense blocks (modules.py:63-95, 474-547 take and return [b, c, D, H, W]):
    a cubic power-of-two grid goes to node rows (b * 8^depth + morton(x, y, z): the layout the whole net runs in) and
    back.  Returns (rows, GridState, back(rows, depth) -> dense)."""
    assert x.dim() == 5 and x.shape[2] == x.shape[3] == x.shape[4], 'dense blocks take cubic [b, c, S, S, S] grids'
    S = x.shape[2]
    depth = S.bit_length() - 1
    assert 1 << depth == S, 'grid edge must be a power of two'
    B = x.shape[0]
    rows = ops.voxel2octree_cf(x.float().contiguous(), depth)
    return rows, GridState(B, depth, x.device), (lambda r, d: ops.octree2voxel_cf(r, B, d))


class GroupNorm32(nn.GroupNorm):
    """reference modules.py:26-28 (parameters ``weight`` / ``bias`` [C]); runs on libofx in row layout."""

    @torch.no_grad()
    def forward(self, x, gs, act=None, out=None):
        return ops.group_norm(x, gs.cache.batch_id(gs.depth), gs.cache.count(gs.depth), gs.B, self.weight,
                              self.bias, self.num_groups, self.eps, act, out, count_eps=0.0,
                              rows_per_batch=8 ** gs.depth)


def convnormalization(channels):
    return GroupNorm32(min(channels, 32), channels)


class our_Identity(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x


class GridConv3d(nn.Module):
    """nn.Conv3d(cin, cout, 3, padding=1[, stride]) parameters; 27-tap gather-GEMM compute.

    mode 0: stride 1; mode 1: stride 2 (depth d -> d-1); mode 2: nearest x2 upsample then conv
    (depth d -> d+1)."""

    def __init__(self, cin, cout, mode=0):
        super().__init__()
        self.in_channels, self.out_channels, self.mode = cin, cout, mode
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(cin * 27)
        nn.init.uniform_(self.bias, -bound, bound)
        self._pw = ops.PackedConv3d()

    def out_depth(self, depth):
        return depth + (0, -1, 1)[self.mode]

    @torch.no_grad()
    def forward(self, x, gs, emb=None, res=None, out=None):
        """``out``: optional destination rows (may be a column slice of a wider buffer: zero-copy concatenation)."""
        d_out = self.out_depth(gs.depth)
        n_out = gs.B * 8 ** d_out
        bid = gs.cache.batch_id(d_out) if emb is not None else None
        mode = self.mode
        return ops.gridconv(x, lambda fast: gs.cache.table(mode, d_out, fast), n_out, self._pw.get(self.weight),
                            self.bias, emb, bid, res, out)


class _PointConv(nn.Module):
    """1x1 conv with nn.Conv{1,3}d-shaped ``weight`` ([cout, cin, 1(,1,1)]) on the MFMA GEMM."""

    def __init__(self, cin, cout, kdims):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, *([1] * kdims)))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.uniform_(self.bias, -1 / math.sqrt(cin), 1 / math.sqrt(cin))
        self._pw = ops.PackedWeight()

    @torch.no_grad()
    def forward(self, x, res=None, out=None):
        w = self.weight.view(self.weight.shape[0], self.weight.shape[1])
        return ops.gemm(x, self._pw.get(w, 'nk'), self.bias, res, out)


class ConvUpsample(nn.Module):
    """reference modules.py:63-78 (``conv.weight``)."""

    def __init__(self, channels, use_conv=True, dims=3):
        super().__init__()
        self.channels = channels
        self.conv = GridConv3d(channels, channels, mode=2)

    def forward(self, x, gs=None, out=None):
        """rows + GridState -> (rows, GridState one level finer); or the reference's ``forward(x)`` on [b, c, D, H, W]."""
        if gs is None:
            rows, gs, back = _dense_io(x)
            return back(self.conv(rows, gs), gs.depth + 1)
        return self.conv(x, gs, out=out), gs.at(gs.depth + 1)


class ConvDownsample(nn.Module):
    """reference modules.py:81-95 (``op.weight``)."""

    def __init__(self, channels, use_conv=True, dims=3):
        super().__init__()
        self.channels = channels
        self.op = GridConv3d(channels, channels, mode=1)

    def forward(self, x, gs=None):
        """rows + GridState -> (rows, GridState one level coarser); or the reference's ``forward(x)`` on [b, c, D, H, W]."""
        if gs is None:
            rows, gs, back = _dense_io(x)
            return back(self.op(rows, gs), gs.depth - 1)
        return self.op(x, gs), gs.at(gs.depth - 1)


class ResnetBlock(nn.Module):
    """reference modules.py:474-513 (use_text_condition False on every live path)."""

    def __init__(self, world_dims, dim_in, dim_out, emb_dim, dropout=0.1, use_text_condition=False):
        super().__init__()
        assert world_dims == 3 and not use_text_condition
        self.time_mlp = nn.Sequential(nn.SiLU(), _Linear(emb_dim, dim_out))
        self.block1 = nn.Sequential(convnormalization(dim_in), nn.SiLU(), GridConv3d(dim_in, dim_out))
        conv2 = GridConv3d(dim_out, dim_out)
        for p in conv2.parameters():
            p.detach().zero_()
        self.block2 = nn.Sequential(convnormalization(dim_out), nn.SiLU(), nn.Dropout(dropout), conv2)
        self.res_conv = _PointConv(dim_in, dim_out, 3) if dim_in != dim_out else nn.Identity()

    @torch.no_grad()
    def forward(self, x, emb_act, gs=None, out=None, t=None):
        """Row layout: emb_act = SiLU(time embedding) [B, emb_dim] (shared by all blocks of a step); ``t``: optional
        precomputed time_mlp(emb) [B, dim_out] (the U-Net evaluates the time_mlp of ALL its blocks in one launch).
        Without ``gs``: the reference's ``forward(x, time_emb)`` (modules.py:505-513) on x [b, c, D, H, W] with the RAW
        time embedding (time_mlp's SiLU is applied here)."""
        if gs is None:
            rows, gs, back = _dense_io(x)
            return back(self.forward(rows, ops.act(emb_act.float().contiguous(), 'silu'), gs), gs.depth)
        # the 1x1 residual convolution only needs x: a parallel branch (ops.fork_stream), joined before conv2 adds it
        skip, main, fork = x, None, None
        if not isinstance(self.res_conv, nn.Identity):
            main, fork = ops.fork_stream(x.device, level=2)
            if fork is not None:
                with torch.cuda.stream(fork):
                    skip = self.res_conv(x)
            else:
                skip = self.res_conv(x)
        h = self.block1[0](x, gs, act='silu')
        if t is None:
            t = self.time_mlp[1](emb_act)                   # [B, dim_out]
        h = self.block1[2](h, gs, emb=t)                    # conv + bias + t[batch] fused
        h = self.block2[0](h, gs, act='silu', out=h)
        if fork is not None:
            main.wait_stream(fork)
        return self.block2[3](h, gs, res=skip, out=out)


class QKVAttention(nn.Module):
    """reference modules.py:538-547 -- computed by ofx_attention."""


class AttentionBlock(nn.Module):
    """reference modules.py:515-535."""

    def __init__(self, channels, num_heads=1):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads
        self.norm = convnormalization(channels)
        self.qkv = _PointConv(channels, channels * 3, 1)
        self.attention = QKVAttention()
        self.proj_out = _PointConv(channels, channels, 1)
        for p in self.proj_out.parameters():
            p.detach().zero_()

    @torch.no_grad()
    def forward(self, x, gs=None, out=None):
        """rows + GridState; or the reference's ``forward(x)`` (modules.py:527-535) on [b, c, D, H, W]."""
        if gs is None:
            rows, gs, back = _dense_io(x)
            return back(self.forward(rows, gs), gs.depth)
        qkv = self.qkv(self.norm(x, gs))
        h = ops.attention(qkv, gs.B, 8 ** gs.depth, self.num_heads)
        return self.proj_out(h, res=x, out=out)


class _AttnSeq(nn.Sequential):
    """Sequential(GroupNorm32, SiLU, AttentionBlock) of graph_unet_lr.py:128-132 (keys 0.*, 2.*)."""

    def forward(self, x, gs, out=None):
        return self[2](self[0](x, gs, act='silu'), gs, out=out)


class LearnedSinusoidalPosEmb(nn.Module):
    """reference modules.py:550-563 ([B] -> [B, dim+1]; a few hundred floats of host-side glue)."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))

    def forward(self, x):
        x = x[:, None]
        freqs = x * self.weights[None, :] * 2 * math.pi
        return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


class UNet3DModel(nn.Module):
    def __init__(self, full_depth, in_split_channels, model_channels, out_split_channels,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), dims=2, num_classes=None,
                 use_checkpoint=False, num_heads=-1, use_text_condition=False, context_dim=None,
                 n_embed=None, **kwargs):
        super().__init__()
        assert dims == 3
        self.full_depth = full_depth
        self.in_channels = in_split_channels
        self.model_channels = model_channels
        self.out_channels = out_split_channels
        self.attention_resolutions = attention_resolutions
        self.channel_mult = list(channel_mult)
        self.num_classes = num_classes
        self.num_heads = num_heads
        mc = model_channels
        chans = [mc] + [mc * m for m in self.channel_mult]
        in_out = list(zip(chans[:-1], chans[1:]))
        ted = mc * 4
        self.time_pos_emb = LearnedSinusoidalPosEmb(mc)
        self.time_emb = nn.Sequential(_Linear(mc + 1, ted), nn.SiLU(), _Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)
        self.input_emb = GridConv3d(2 * self.in_channels, mc)

        def attn(c, ds):
            if ds in attention_resolutions:
                return _AttnSeq(convnormalization(c), nn.SiLU(), AttentionBlock(c, num_heads=num_heads))
            return our_Identity()

        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        nres = len(in_out)
        ds = 1
        for ind, (ci, co) in enumerate(in_out):
            last = ind >= nres - 1
            self.downs.append(nn.ModuleList([
                ResnetBlock(dims, ci, co, emb_dim=ted, dropout=dropout), attn(co, ds),
                ConvDownsample(co, dims=dims) if not last else our_Identity()]))
            if not last:
                ds *= 2
        mid = chans[-1]
        self.mid_block1 = ResnetBlock(dims, mid, mid, emb_dim=ted, dropout=dropout)
        self.mid_self_attn = attn(mid, ds)
        self.mid_block2 = ResnetBlock(dims, mid, mid, emb_dim=ted, dropout=dropout)
        # graph_unet_lr.py:152-166: walks reversed(in_out[1:]); every level upsamples
        for ci, co in reversed(in_out[1:]):
            self.ups.append(nn.ModuleList([
                ResnetBlock(dims, co * 2, ci, emb_dim=ted, dropout=dropout), attn(ci, ds),
                ConvUpsample(ci, dims=dims)]))
            ds //= 2
        self.end = nn.Sequential(convnormalization(mc), nn.SiLU())
        self.out = GridConv3d(mc, self.out_channels)

    # ---- row-layout core ------------------------------------------------
    @torch.no_grad()
    def _embed(self, timesteps, label, batch_size):
        """SiLU(time_emb(time_pos_emb(t)) + label_emb(label)) [B, 4 mc] (graph_unet_lr.py:186-193): three launches."""
        pe = ops.learned_sinusoid(timesteps.float(), self.time_pos_emb.weights)
        lab = None
        if self.num_classes is not None:
            assert label.shape == (batch_size,)
            lab = self.label_emb(label)
        return self.time_emb[2](self.time_emb[0](pe, act_out='silu'), res=lab, act_out='silu')

    def _all_time_mlps(self, emb_act):
        """{id(block): time_mlp(emb) [B, dim_out]} of every ResnetBlock from ONE launch on the row-concatenated
        weights (every block applies its own Linear to the same SiLU(emb), modules.py:507-511)."""
        blocks = [m for m in self.modules() if isinstance(m, ResnetBlock)]
        lins = [b.time_mlp[1] for b in blocks]
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version) for l in lins)
        if getattr(self, '_tm_key', None) != key:
            self._tm_w = torch.cat([l.weight.detach() for l in lins], dim=0).contiguous()
            self._tm_b = torch.cat([l.bias.detach() for l in lins], dim=0).contiguous()
            self._tm_pw = ops.PackedWeight()
            self._tm_key = key
        if ops.LINEAR_SMALL and emb_act.shape[0] <= 16:
            allout = ops.linear_small(emb_act, self._tm_w, self._tm_b)
        else:
            allout = ops.gemm(emb_act, self._tm_pw.get(self._tm_w, 'nk'), self._tm_b)
        outs, off = {}, 0
        for b, l in zip(blocks, lins):
            outs[id(b)] = allout[:, off:off + l.out_features]
            off += l.out_features
        return outs

    @torch.no_grad()
    def forward_rows(self, x, batch_size, timesteps, label=None, as_middle=False, out=None):
        """x [B*8^full_depth, C] in node-row layout; returns rows at full_depth.  (ops.POLICY['dense_net'] can move this
        net to another contraction mode when the global mode is 'bf16x3' -- an instrument of the precision study.)"""
        with ops.policy_scope('dense_net'), ops.stats_scope(x.device):
            return self._forward_rows(x, batch_size, timesteps, label, as_middle, out)

    @torch.no_grad()
    def precompute_embeddings(self, timesteps, label, batch_size):
        """The embedding chain (4 launches that depend on the timesteps only) on the fork stream; the nesting sparse net
        calls this at the START of its own forward, so that the chain runs beside its encoder instead of in the middle of
        the step.  Consumed (and joined) by the next _forward_rows with the same timesteps tensor."""
        main, fork = ops.fork_stream(timesteps.device)
        if fork is None:
            return
        with torch.cuda.stream(fork):
            emb_act = self._embed(timesteps, label, batch_size)
            tms = self._all_time_mlps(emb_act)
        self._pre = (timesteps, label, batch_size, emb_act, tms, main, fork)

    def _forward_rows(self, x, batch_size, timesteps, label, as_middle, out=None):
        gs = GridState(batch_size, self.full_depth, x.device)
        if not as_middle:
            x = self.input_emb(x, gs)
        pre, self._pre = getattr(self, '_pre', None), None
        if pre is not None and pre[0] is timesteps and pre[1] is label and pre[2] == batch_size:
            emb_act, tms = pre[3], pre[4]
            pre[5].wait_stream(pre[6])
        else:
            if pre is not None:
                pre[5].wait_stream(pre[6])              # (an unused precompute: still joined, never left dangling)
            emb_act = self._embed(timesteps, label, batch_size)
            tms = self._all_time_mlps(emb_act)

        # Zero-copy skip concatenation (as the sparse net does): the decoder level that consumes the skip tensor of
        # encoder level i reads ONE buffer [rows_i, C_x + C_skip]; the module that produces the skip writes its right
        # columns and the module that produces x (mid_block2 / the previous level's upsampling conv) its left columns,
        # so torch.cat (graph_unet_lr.py:210) never copies.  Level 0's skip is never consumed (the reference walks
        # reversed(in_out[1:]), :152).
        nlev = len(self.downs)
        mc = self.model_channels
        cout = [mc * m_ for m_ in self.channel_mult]                   # channels leaving encoder level i
        bufs = {}
        for i in range(1, nlev):
            d_i = self.full_depth - min(i, nlev - 1)
            bufs[i] = torch.empty(batch_size * 8 ** d_i, 2 * cout[i], dtype=torch.float32, device=x.device)

        def run_attn(m, x, gs, out=None):
            return x if isinstance(m, our_Identity) else m(x, gs, out=out)

        for i, (resnet, self_attn, downsample) in enumerate(self.downs):
            slot = bufs[i][:, cout[i]:] if i in bufs else None
            if isinstance(self_attn, our_Identity):
                x = resnet(x, emb_act, gs, out=slot, t=tms[id(resnet)])
            else:
                x = run_attn(self_attn, resnet(x, emb_act, gs, t=tms[id(resnet)]), gs, out=slot)
            if not isinstance(downsample, our_Identity):
                x, gs = downsample(x, gs)
        x = self.mid_block1(x, emb_act, gs, t=tms[id(self.mid_block1)])
        x = run_attn(self.mid_self_attn, x, gs)
        x = self.mid_block2(x, emb_act, gs, out=bufs[nlev - 1][:, :cout[nlev - 1]] if nlev > 1 else None,
                            t=tms[id(self.mid_block2)])
        for j, (resnet, self_attn, upsample) in enumerate(self.ups):
            i = nlev - 1 - j                                            # the encoder level whose skip this level consumes
            x = bufs[i]                                                 # = cat((x, skip_i), dim=1)
            x = run_attn(self_attn, resnet(x, emb_act, gs, t=tms[id(resnet)]), gs)
            nxt = bufs[i - 1][:, :cout[i - 1]] if i - 1 in bufs else None
            x, gs = upsample(x, gs, out=nxt)
        x = self.end[0](x, gs, act='silu', out=out if as_middle else None)
        return x if as_middle else self.out(x, gs)

    # ---- reference signatures ---------------------------------------------
    @torch.no_grad()
    def forward_as_middle(self, h, doctree, timesteps, label, context, out=None):
        # rows of the full layer ARE the dense grid: no octree2voxel / gather-back needed.  ``out``: optional destination
        # rows (a column slice of the caller's concatenation buffer -- the sparse net's cat([h, h_lr]) without a copy)
        return self.forward_rows(h, doctree.batch_size, timesteps, label, as_middle=True, out=out)

    @torch.no_grad()
    def forward(self, x=None, timesteps=None, x_self_cond=None, label=None, context=None, as_middle=False,
                **kwargs):
        """Dense-tensor interface of the reference: x [B, C, S, S, S] -> [B, C', S, S, S]."""
        assert (label is not None) == (self.num_classes is not None), \
            'must specify label if and only if the model is class-conditional'
        B = x.shape[0]
        if not as_middle:
            # cat((x, x_self_cond), dim=1) in row layout: both halves are written straight into one rows buffer
            C = x.shape[1]
            rows = torch.empty(B * 8 ** self.full_depth, 2 * C, dtype=torch.float32, device=x.device)
            ops.voxel2octree_cf(x.float(), self.full_depth, out=rows[:, :C])
            if x_self_cond is None:
                rows[:, C:].zero_()
            else:
                ops.voxel2octree_cf(x_self_cond.float(), self.full_depth, out=rows[:, C:])
        else:
            rows = ops.voxel2octree_cf(x.float(), self.full_depth)
        y = self.forward_rows(rows, B, timesteps, label, as_middle)
        return ops.octree2voxel_cf(y, B, self.full_depth)
