"""DDIM sampling driver: the loop that calls the U-Net once per denoising step.

Own code reproducing the semantics of reference
models/octfusion_model_union.py:293-352 (``sample_loop``): continuous-time
log-SNR schedule (ldm_diffusion_util.py:293-309), x0 branch with the 0.7
truncation + sign_() for the "lr" stage, eps branch for "hr" / "feature".  The
elementwise updates are libofx kernels whose coefficients live in device memory
(so a step can be replayed from a hipGraph); the per-step scalars are computed
on the host in fp32 exactly as the reference computes them.
"""
import os

import torch

from . import ops

TRUNCATED_TIME = 0.7      # octfusion_model_union.py:39


def beta_linear_log_snr(t):
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def sampling_times(steps):
    """[(t, t_next)] for linspace(1, 0, steps+1) (octfusion_model_union.py:293-298)."""
    times = torch.linspace(1., 0., steps + 1)
    return [(times[i], times[i + 1]) for i in range(steps)]


def eps_coef(t, t_next):
    a, s = log_snr_to_alpha_sigma(beta_linear_log_snr(t))
    an, sn = log_snr_to_alpha_sigma(beta_linear_log_snr(t_next))
    return torch.stack([a, s, an, sn]).float()


def x0_coef(t, t_next, truncated_index):
    ls, lsn = beta_linear_log_snr(t), beta_linear_log_snr(t_next)
    a, s = log_snr_to_alpha_sigma(ls)
    an, sn = log_snr_to_alpha_sigma(lsn)
    c = -torch.special.expm1(ls - lsn)
    sd = torch.sqrt((sn ** 2) * c) if bool(t_next > truncated_index) else torch.zeros(())
    return torch.stack([a, c, an, sd]).float()


def ddim_eps_step(x, eps, t, t_next):
    coef = eps_coef(torch.as_tensor(t, dtype=torch.float32), torch.as_tensor(t_next, dtype=torch.float32))
    return ops.ddim_eps_update(x, eps.contiguous(), coef.to(x.device))


@torch.no_grad()
def _step(net, x, noise_cond, unet_type, df_type, doctree, unet_lr, label, x_self, coef, noise, do_sign, x0_out):
    """One denoising step on device tensors only (what gets captured into a hipGraph): U-Net forward + update.
    Returns the tensor the next step receives as x_self_cond."""
    out = net(unet_type=unet_type, x=x, doctree=doctree, timesteps=noise_cond, unet_lr=unet_lr,
              x_self_cond=x_self, label=label)
    if do_sign:
        out = out.sign_()
    out = out.contiguous()
    if df_type == 'x0':
        ops.ddim_x0_update(x, out, noise, coef)
        return out
    if df_type == 'eps':
        # the reference also keeps x_start = (x - eps*sigma)/alpha and passes it on as x_self_cond
        # (octfusion_model_union.py:349, :320); the hr / feature nets ignore it, so it is only
        # materialised for nets that declare they want it.
        ops.ddim_eps_update(x, out, coef, x0_out)
        return x0_out
    raise ValueError(df_type)


LANES = int(os.environ.get('OFX_LANES', '1'))         # lanes of a graph stage (hr / feature); 1 = the whole batch on one stream
LANE_MIN_BATCH = 2
LANE_CUS = int(os.environ.get('OFX_LANE_CUS', '128'))  # compute units a persistent GraphConv launch of a lane is planned for (0: all)
_LANE_STREAMS = {}


def lane_streams(device, n):
    """The n HIP streams of `device` the lanes of sample_loop run on (created once: per-stream scratch is keyed by them)."""
    ss = _LANE_STREAMS.setdefault(device.index, [])
    while len(ss) < n:
        ss.append(torch.cuda.Stream(device))
    return ss[:n]


def lane_count(batch_size, doctree, use_graph):
    """How many lanes a stage runs as (OFX_LANES, default 1): graph stages of at least LANE_MIN_BATCH shapes, replayed
    from hipGraphs, can run as LANES runs of consecutive shapes on separate HIP streams -- while one lane sits in a
    matrix-bound GraphConv the other is in an HBM-bound GroupNorm / gather launch.  Measured on MI355X, same box, bench.py
    --lanes 1 / 2 (profiles/r06/lanes_ab.txt): hr B = 8 8.34 -> 8.16 ms (-2 %), hr_cond B = 4 6.65 -> 6.29 (-5 %), feature
    42.7 -> 42.0 (-2 %); the dense lr stage loses 3 %, four lanes lose to two.  The half-batch launches are individually
    less efficient (the 128 -> 128 GraphConv alone: 0.36 of the matrix roof at half the rows against 0.41), which eats
    most of the overlap: a few per cent for twice the graphs and scratch, so it stays opt-in."""
    if not use_graph or doctree is None or LANES < 2 or batch_size < max(LANE_MIN_BATCH, 2):
        return 1
    if not hasattr(doctree, 'split_batch'):
        return 1
    return min(LANES, batch_size)


def sample_loop(net, shape, batch_size, ddim_steps, unet_type, df_type, device, doctree=None, unet_lr=None,
                label=None, truncated_index=0.0, init_noise=None, step_noise=None, use_graph=None, lanes=None):
    """Run `ddim_steps` denoising steps; `net` is a graph_unet_union.UNet3DModel (or any callable with
    its keyword interface).  Noise comes from torch's device RNG unless given explicitly.

    use_graph (default: on for HIP devices): every shape is static across the steps of a stage (the doctree is
    fixed), so after one eager step the step is captured into a hipGraph per regime (sign / noise / first-step
    flags) and replayed with x, log-SNR, coefficients, noise and the self-conditioning tensor in static device
    buffers -- ~250 launches per step leave the host's critical path.

    lanes (default lane_count()): a graph stage's batch is split into that many runs of consecutive shapes
    (DualOctree.split_batch), each with its own rows of x, static buffers, hipGraphs and HIP stream; the host issues
    step i of every lane before step i + 1 and the streams never wait for each other until the last step.  The initial
    noise (and an explicit `step_noise`) is that of the whole batch, split by rows; per-step noise from the device RNG
    (x0 stages) is drawn per lane -- the same distribution, not the same stream of numbers as a one-lane call.
    Results equal the one-lane call's up to the summation order inside a GraphConv launch (its k-split follows the
    row count): tests/test_gpu_lanes.py."""
    if use_graph is None:
        use_graph = torch.device(device).type == 'cuda' and isinstance(net, torch.nn.Module)
    x = torch.randn(shape, device=device) if init_noise is None else init_noise.to(device).clone()
    x = x.contiguous()
    if x.is_cuda:
        ops.reset_range_words(x.device)
    n_lanes = lane_count(batch_size, doctree, use_graph) if lanes is None else int(lanes)
    parts = None
    if n_lanes > 1:
        parts = doctree.split_batch(n_lanes)
        if len(parts) < 2 or any(int(p[0].nnum[d]) == 0 for p in parts for d in range(p[0].full_depth, p[0].depth + 1)):
            parts = None                     # a part without nodes at some depth: the whole batch on one stream
    if parts is None:
        for _ in _lane_steps(net, x, batch_size, ddim_steps, unet_type, df_type, device, doctree, unet_lr, label,
                             truncated_index, step_noise, use_graph, None):
            pass
    else:
        main = torch.cuda.current_stream(x.device)
        streams = lane_streams(x.device, len(parts))
        lanes_ = []
        for (sub, rows, (b0, b1)), s in zip(parts, streams):
            xs = x.index_select(0, rows)
            noise_p = None if step_noise is None else [None if n is None else n.to(device).index_select(0, rows) for n in step_noise]
            s.wait_stream(main)
            lanes_.append((s, rows, xs, _lane_steps(net, xs, b1 - b0, ddim_steps, unet_type, df_type, device, sub, unet_lr,
                                                    None if label is None else label[b0:b1].contiguous(), truncated_index,
                                                    noise_p, use_graph, s)))
        ops.set_lane_cus(LANE_CUS)           # persistent launches planned so that two lanes' launches are resident side by side
        try:
            for i in range(ddim_steps):
                for j, (s, _, _, gen) in enumerate(lanes_):
                    if i == 0 and j > 0:
                        # the first step packs the net's weights lazily, on the FIRST lane's stream: the other lanes read
                        # the same packs, so their first step waits for it (nothing else is shared between lanes)
                        s.wait_stream(lanes_[0][0])
                    with torch.cuda.stream(s):
                        next(gen)
        finally:
            ops.set_lane_cus(0)
        for s, rows, xs, gen in lanes_:
            gen.close()
            main.wait_stream(s)
            x.index_copy_(0, rows, xs)
    if x.is_cuda:
        ops.raise_on_sync_error(x.device)       # (one host read per stage; the caller synchronises right after anyway)
        ops.raise_on_range_error(x.device, x)
    return x


def _lane_steps(net, x, batch_size, ddim_steps, unet_type, df_type, device, doctree, unet_lr, label, truncated_index,
                step_noise, use_graph, capture_stream):
    """Generator: one denoising step of one lane per next(), on the CURRENT stream, updating x in place.
    capture_stream: the stream warm-up and capture of this lane's hipGraphs run on (None: the device's side stream;
    a lane passes its own stream -- scratch is kept per stream, so lanes must not share it)."""
    wants_sc = getattr(net, 'wants_self_cond', True)
    x_start = None
    graphs = {}
    st = {}
    if use_graph:
        st['cond'] = torch.zeros(batch_size, dtype=torch.float32, device=device)
        st['self'] = torch.zeros_like(x)
        st['noise'] = torch.zeros_like(x)
    times = sampling_times(ddim_steps)
    # every per-step scalar is computed on the host in fp32 (as the reference does) and uploaded ONCE:
    # one [steps, 4] coefficient table and one [steps] log-SNR vector, indexed per step (no per-step H2D / D2H)
    coef_host = torch.stack([x0_coef(t, tn, truncated_index) if df_type == 'x0' else eps_coef(t, tn)
                             for t, tn in times])
    coef_dev = coef_host.to(device)
    cond_dev = torch.stack([beta_linear_log_snr(t).float() for t, _ in times]).to(device)
    for i, (t, t_next) in enumerate(times):
        noise_cond = cond_dev[i].expand(batch_size).contiguous()
        # `t[0] < truncated_index` in the reference compares an fp32 tensor with a scalar IN fp32
        # (octfusion_model_union.py:335): linspace(1, 0, steps+1) contains fp32(0.7) = 0.69999999 for
        # steps = 10, 20, 50, 100, 200, where a double-precision compare would sign one step early
        do_sign = bool(t < truncated_index) and unet_type == 'lr'
        coef = coef_dev[i]
        noise = None
        if df_type == 'x0' and float(coef_host[i, 3]) != 0.0:
            noise = torch.randn_like(x) if step_noise is None else step_noise[i].to(device)
        if not use_graph or i == 0:
            x0_out = torch.empty_like(x) if (df_type == 'eps' and wants_sc) else None
            x_start = _step(net, x, noise_cond, unet_type, df_type, doctree, unet_lr, label, x_start, coef, noise,
                            do_sign, x0_out)
            yield i
            continue
        # ---- hipGraph replay -------------------------------------------------------------------------
        key = (do_sign, noise is not None, x_start is not None)
        st['cond'].copy_(noise_cond)
        if x_start is not None and x_start is not st['self']:
            st['self'].copy_(x_start)
        if noise is not None:
            st['noise'].copy_(noise)
        if key not in graphs:
            c = coef.clone()
            x0_buf = torch.empty_like(x) if (df_type == 'eps' and wants_sc) else None
            args = (net, x, st['cond'], unet_type, df_type, doctree, unet_lr, label,
                    st['self'] if key[2] else None, c, st['noise'] if key[1] else None, do_sign, x0_buf)
            g = torch.cuda.CUDAGraph()
            keep = x.clone()
            side = capture_stream or ops.side_stream(x.device)   # ONE warm-up stream per device / lane: scratch is kept per stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _step(*args)                           # warm-up of this regime outside the capture
            torch.cuda.current_stream().wait_stream(side)
            x.copy_(keep)
            # capture on the warm-up stream: its flag words and workspace (ops.sync_words / ops.workspace are per stream)
            # were allocated by the eager warm-up above, so the capture contains no allocation and no zero-fill of the
            # sticky error word
            with torch.cuda.graph(g, stream=side):
                res = _step(*args)
            x.copy_(keep)
            graphs[key] = (g, c, res)
        g, c, res = graphs[key]
        c.copy_(coef)
        g.replay()
        x_start = res
        yield i
