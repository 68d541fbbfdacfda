#!/usr/bin/env python
"""bench.py -- denoising-steps/sec of OctFusion's diffusion U-Nets on MI355X.

Default workload = BASELINE.json configs[2], the one `metric` is quoted on: ShapeNet uncond "hr" stage
(configs/octfusion_snet_uncond.yaml) on the synthetic shell-6 octree batch (diffusion depth 6 of the
depth-8 VAE octree), batch 8 per GPU, DDIM eps-branch.  One step = one full U-Net forward (sparse hr net + the
nested dense lr net) + the DDIM update for the whole batch.  Inputs are resident in HBM before the timed region.
fp32 storage; default contraction fp16x3 (fp16 hi/lo operand pairs, three fp16 MFMAs per product, fp32 accumulate:
the fp32 reference's rounding class); exact fp32 MFMA and the bf16-pair variant are timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload hr|lr|hr_cond|feature]

`--gpus N` (N > 1) starts N ranks by itself (re-executes under torch.distributed.run on 127.0.0.1) unless it is
already running under a launcher (WORLD_SIZE set).  Prints ONE JSON line on rank 0 (DESIGN.md "Measurement").
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_32x32x2_f32)
MFMA_16BIT_PEAK_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (no 2:1 sparsity)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak
# matrix roof of the ALGORITHMIC flops per contraction mode: the three-term modes issue 3 MFMAs per product
PEAKS = {'fp16x3': MFMA_16BIT_PEAK_TFLOPS / 3.0, 'bf16x3': MFMA_16BIT_PEAK_TFLOPS / 3.0, 'fp32': MFMA_F32_PEAK_TFLOPS,
         'fp16': MFMA_16BIT_PEAK_TFLOPS}

# BASELINE.json configs -> concrete synthetic inputs
WORKLOADS = {
    'hr': dict(config='snet_uncond', stage='hr', tree='shell6', batch=8, df='eps', label=False, steps=300,
               cpu=(3, 5), desc='BASELINE configs[2]: snet_uncond stage hr (+nested lr), shell-6 octree (diffusion '
                                'depth 6 of the depth-8 VAE octree), batch %d per GPU, DDIM eps step'),
    'lr': dict(config='snet_uncond', stage='lr', tree=None, batch=4, df='x0', label=False, steps=1000,
               cpu=(3, 5), desc='BASELINE configs[1]: snet_uncond stage lr (dense 16^3 net with attention), '
                                'batch %d per GPU, DDIM x0 step with self-conditioning'),
    'hr_cond': dict(config='snet_cond', stage='hr', tree='shell6', batch=4, df='eps', label=True, steps=400,
                    cpu=(3, 5), desc='BASELINE configs[3] per-GPU slice: snet_cond stage hr (+nested lr), 5 classes '
                                     '(labels b mod 5), shell-6 octree, batch %d per GPU, DDIM eps step'),
    'feature': dict(config='obja_uncond', stage='feature', tree='shell8', batch=8, df='x0', label=False, steps=30,
                    cpu=(1, 2), desc='BASELINE configs[4] per-GPU slice: obja_uncond stage feature (hr nested as its '
                                     'middle), shell-8 octree (N8 = 448 232 per shape), batch %d per GPU, DDIM x0 step'),
}
NESTED = {'hr': 'unet_lr', 'feature': 'unet_hr', 'lr': None}


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_bootstrap_cmd(argv, gpus, port=None):
    """The command `python bench.py --gpus N` re-executes itself as when no launcher started it (one rank per GPU)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port or free_port()), os.path.abspath(__file__)] + argv


def build_tree(kind, batch, dev):
    """(octree, doctree, setup_ms) of the synthetic workload tree; setup = octree build + dual-graph build."""
    from octfusion_amd import synthetic
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_large, split2octree_small
    split = synthetic.shell6_split(batch, jitter=True).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    oc = split2octree_small(split, 6, 4)
    if kind == 'shell8':
        x6, y6, z6, _ = oc.xyzb(6)
        sl = synthetic.shell8_split_large(x6, y6, z6)
        oc = split2octree_large(oc, sl, 6)
    doc = DualOctree(oc)
    torch.cuda.synchronize()
    return oc, doc, 1e3 * (time.perf_counter() - t0)


class Workload:
    """One denoising step of a stage, driven exactly like sampler.sample_loop drives it."""

    def __init__(self, name, batch, dev, rank, lanes=None):
        from octfusion_amd import configs, dist, sampler, synthetic
        from octfusion_amd.graph_unet_union import UNet3DModel
        w = WORKLOADS[name]
        self.name, self.w, self.batch, self.dev = name, w, batch, dev
        self.stage, self.df = w['stage'], w['df']
        self.net = UNet3DModel(**configs.unet_params(w['config'], self.stage))
        if rank == 0:
            self.net.load_state_dict(synthetic.random_state_dict(self.net))
        self.net = self.net.to(dev).eval()
        self.bcast_bytes = dist.broadcast_module_(self.net, src=0)        # the only collective: weights, once
        self.doc = None
        self.setup_ms = self.setup_warm_ms = 0.0
        if w['tree']:
            _, self.doc, self.setup_ms = build_tree(w['tree'], batch, dev)          # first call: includes lazy code loading
            self.setup_warm_ms = build_tree(w['tree'], batch, dev)[2]               # what every later batch of shapes pays
            self.shape = (self.doc.total_num, configs.CONFIGS[w['config']]['input_channels'][-1 if self.stage == 'feature' else 1])
        else:
            self.shape = (batch, 8, 16, 16, 16)
        g = torch.Generator().manual_seed(1 + rank)
        self.x = torch.randn(self.shape, generator=g).to(dev)
        self.x_init = self.x.clone()
        self.label = (torch.arange(batch) % 5).to(dev) if w['label'] else None
        self.nested = getattr(self.net, NESTED[self.stage]) if NESTED[self.stage] else None
        times = sampler.sampling_times(200)
        trunc = sampler.TRUNCATED_TIME if self.stage == 'lr' else 0.0
        self.coef_host = torch.stack([sampler.x0_coef(t, tn, trunc) if self.df == 'x0' else sampler.eps_coef(t, tn)
                                      for t, tn in times])
        self.coef = self.coef_host.to(dev)
        self.cond = torch.stack([sampler.beta_linear_log_snr(t).float() for t, _ in times]).to(dev)
        self.sign = [bool(t < trunc) and self.stage == 'lr' for t, _ in times]
        self.x_self = None
        self.sampler = sampler
        # graph stages run as sampler.sample_loop runs them: LANES runs of consecutive shapes, each on its own HIP stream
        self.lanes, self.serial, self.lane_split_ms = [], False, 0.0
        self.n_lanes = sampler.lane_count(batch, self.doc, True) if lanes is None else int(lanes)
        if self.doc is not None and self.n_lanes > 1:
            self.make_lanes()

    RESET_EVERY = 50

    def make_lanes(self):
        """Split self.doc into the lanes of sampler.sample_loop (DualOctree.split_batch): per lane its doctree, its rows of
        x, its labels and its stream."""
        import types
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parts = self.doc.split_batch(self.n_lanes)
        torch.cuda.synchronize()
        self.lane_split_ms = 1e3 * (time.perf_counter() - t0)
        self.lanes = []
        for (sub, rows, (b0, b1)), s in zip(parts, self.sampler.lane_streams(self.dev, len(parts))):
            L = types.SimpleNamespace(doc=sub, rows=rows, batch=b1 - b0, stream=s, x=self.x_init.index_select(0, rows),
                                      label=None if self.label is None else self.label[b0:b1].contiguous())
            L.x_init = L.x.clone()
            self.lanes.append(L)
        torch.cuda.synchronize()

    def set_doc(self, doc):
        self.doc = doc
        if self.lanes:
            self.make_lanes()

    def lane_order(self):
        """(lane, stream to wait for first or None): lanes overlap freely (the timed region) unless self.serial, where
        every lane's step starts after the previous lane's has finished (per-launch event timings: a kernel alone on the chip)."""
        prev = self.lanes[-1] if self.serial else None
        for L in self.lanes:
            yield L, (prev.stream if prev is not None and prev is not L else None)
            prev = L if self.serial else None

    def finite(self):
        return all(bool(torch.isfinite(L.x).all()) for L in self.lanes) if self.lanes else bool(torch.isfinite(self.x).all())

    def reset_input(self, i):
        """With random weights the DDIM update is not a denoiser: |x| grows a few per cent per step (hr_cond: x 1.065),
        and after a few hundred steps the residual stream leaves the range real sampling lives in (and the fp16 operand
        range: NaN by design, include/ofx.h).  The state is therefore re-drawn every RESET_EVERY steps -- one N x 3 copy,
        outside the K-step region the driver times (steps W+1 .. W+K with K <= 50 never cross a reset)."""
        if i % self.RESET_EVERY == 0 and i > 0:
            self.x.copy_(self.x_init)
            if self.x_self is not None:
                self.x_self = None

    def step(self, i):
        if self.lanes:
            j = i % 200
            for L, after in self.lane_order():
                if after is not None:
                    L.stream.wait_stream(after)
                with torch.cuda.stream(L.stream):
                    if i % self.RESET_EVERY == 0 and i > 0:
                        L.x.copy_(L.x_init)
                    noise = torch.randn_like(L.x) if self.df == 'x0' and float(self.coef_host[j, 3]) != 0.0 else None
                    self.sampler._step(self.net, L.x, self.cond[j].expand(L.batch).contiguous(), self.stage, self.df, L.doc,
                                       self.nested, L.label, None, self.coef[j], noise, False, None)
            return
        self.reset_input(i)
        i = i % 200
        noise = None
        if self.df == 'x0' and float(self.coef_host[i, 3]) != 0.0:
            noise = torch.randn_like(self.x)
        cond = self.cond[i].expand(self.batch).contiguous()
        self.x_self = self.sampler._step(self.net, self.x, cond, self.stage, self.df, self.doc, self.nested, self.label,
                                         self.x_self if self.stage == 'lr' else None, self.coef[i], noise,
                                         self.sign[i], None)

    def run(self, first, n):
        for i in range(first, first + n):
            self.step(i)


def capture_lanes(wl):
    """One hipGraph per lane, captured on the lane's stream after two eager steps on it (its per-stream scratch exists then)."""
    for L in wl.lanes:
        L.cond_s = wl.cond[0].expand(L.batch).contiguous().clone()
        L.coef_s = wl.coef[0].clone()
        L.noise_s = torch.randn_like(L.x) if wl.df == 'x0' else None

        def lstep(L=L):
            return wl.sampler._step(wl.net, L.x, L.cond_s, wl.stage, wl.df, L.doc, wl.nested, L.label, None,
                                    L.coef_s, L.noise_s, False, None)
        with torch.cuda.stream(L.stream):
            lstep()
            lstep()
            L.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(L.graph, stream=L.stream):
                lstep()


def replay_lanes(wl, first, n):
    for i in range(first, first + n):
        j = i % 200
        for L in wl.lanes:
            with torch.cuda.stream(L.stream):
                if i % wl.RESET_EVERY == 0 and i > 0:
                    L.x.copy_(L.x_init)
                L.cond_s.copy_(wl.cond[j].expand(L.batch))
                L.coef_s.copy_(wl.coef[j])
                if L.noise_s is not None:
                    L.noise_s.normal_()
                L.graph.replay()


def timed(fn, n_sync=True):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota), or None if unlimited.
    The GPU boxes of this project show 256 logical CPUs and a quota of 16: every CPU figure has to live inside it."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def _cpu_stepper(name, batch):
    """(step(i, x) -> x, x0, shapes per step) of the reference's CPU path restated (oracle/) for workload `name`."""
    from octfusion_amd import configs, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU
    w = WORKLOADS[name]
    net = UNet3DModel(**configs.unet_params(w['config'], w['stage']))
    sd = synthetic.random_state_dict(net)
    st = configs.stage_cfgs(w['config'])
    g = torch.Generator().manual_seed(1)
    times = OS.get_sampling_timesteps(1, 200)
    stage = w['stage']
    if stage == 'lr':
        bs = batch
        x = torch.randn(bs, 8, 16, 16, 16, generator=g)
        part = OM._sub(sd, 'unet_lr')

        def fwd(x, ls):
            return OU.lr_forward(part, st['lr'], x, ls.expand(bs), torch.zeros_like(x), None)
    else:
        bs = 1
        oc = OS.split2octree_small(synthetic.shell6_split(1, jitter=False), 6, 4)
        if w['tree'] == 'shell8':
            x6, y6, z6, _ = oc.xyzb(6)
            oc = OS.split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
        doc = OD.OracleDualOctree(oc)
        doc.post_processing_for_docnn()
        x = torch.randn(doc.total_num, 3, generator=g)
        outer, nested = ('unet_hr', 'unet_lr') if stage == 'hr' else ('unet_feature', 'unet_hr')
        po, pn = OM._sub(sd, outer), OM._sub(sd, nested)
        label = torch.zeros(1, dtype=torch.long) if w['label'] else None

        def fwd(x, ls):
            return OU.hr_forward(po, st[outer[5:]], x, doc, ls.expand(1), label, pn, st[nested[5:]])

    def step(i, x):
        t, tn = times[i % len(times)]
        ls, lsn = OS.beta_linear_log_snr(t), OS.beta_linear_log_snr(tn)
        out = fwd(x, ls)
        a, s = OS.log_snr_to_alpha_sigma(ls)
        an, sn = OS.log_snr_to_alpha_sigma(lsn)
        if w['df'] == 'eps':
            x0 = (x - out * s[0]) / a[0].clamp(min=1e-8)
            return x0 * an[0] + out * sn[0]
        return out * an[0] + (x - a[0] * out) / s[0].clamp(min=1e-8) * sn[0]          # deterministic x0-branch update
    return step, x, bs


def _cpu_time_steps(name, batch, n_warm, n_timed, threads, sync_dir=None, k=0, n_workers=1):
    """n_warm + n_timed oracle steps with `threads` torch threads; with sync_dir the timed part starts when all
    n_workers workers have finished their warm-up (file rendezvous)."""
    torch.set_num_threads(threads)
    step, x, bs = _cpu_stepper(name, batch)
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(n_warm):
            x = step(i, x)
        tw = time.perf_counter() - t0
        if sync_dir:
            open(os.path.join(sync_dir, 'ready_%d' % k), 'w').close()
            t_wait = time.time()
            while len([f for f in os.listdir(sync_dir) if f.startswith('ready_')]) < n_workers and time.time() - t_wait < 300:
                time.sleep(0.01)
        w0 = time.time()
        t0 = time.perf_counter()
        for i in range(n_timed):
            x = step(n_warm + i, x)
        dt = time.perf_counter() - t0
    return {'dt': dt, 'warm': tw, 'shapes_per_step': bs, 'wall_start': w0, 'wall_end': w0 + dt}


def cpu_worker_main(argv):
    """`bench.py --cpu-worker name batch k n_workers cpus_per_worker n_warm n_timed sync_dir`: one pinned worker of the
    whole-box CPU baseline."""
    name, batch, k, n_workers, per, n_warm, n_timed = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5]), int(argv[6])
    try:
        allowed = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, set(allowed[k * per:(k + 1) * per]))
    except (AttributeError, OSError):
        pass
    print(json.dumps(_cpu_time_steps(name, batch, n_warm, n_timed, per, argv[7], k, n_workers)))


def cpu_baseline(name, batch):
    """The reference's CPU path restated (oracle/), timed on this host's cores on a bounded sample of the same
    workload (fp32, warm-ups then timed steps; B = 1 shape per step for the sparse stages,
    scaled to the per-GPU batch by division).  Two figures:
      * one process, 32 torch threads (torch-CPU scatter / index ops collapse when one process is given all threads:
        140 s per hr step with 256 threads in round 1, 1.3 s with 32);
      * the WHOLE BOX when it has more than 64 USABLE CPUs (affinity and cgroup quota): usable // 32 processes, each
        pinned to its own 32 logical CPUs, one shape each, running concurrently -- `value` is the better of the two."""
    import tempfile
    w = WORKLOADS[name]
    ncpu = os.cpu_count() or 1
    try:
        ncpu_allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        ncpu_allowed = ncpu
    quota = cgroup_cpu_quota()
    usable = ncpu_allowed if quota is None else max(1, min(ncpu_allowed, int(quota + 0.5)))
    threads = min(32, usable)
    n_warm, n_timed = w['cpu']
    r1 = _cpu_time_steps(name, batch, n_warm, n_timed, threads)
    single = r1['shapes_per_step'] * n_timed / r1['dt'] / batch
    res = {'value': single, 'unit': 'denoising-steps/sec (batch %d)' % batch, 'cores': threads,
           'kind': 'port', 'host_cpu_count': ncpu, 'cgroup_cpu_quota': quota, 'usable_cpus': usable,
           'cpu_model': cpu_model(), 'threads_used': threads,
           'single_process': {'value': single, 'threads': threads, 'timed_s': r1['dt'], 'warmup_s': r1['warm']},
           'sample': 'oracle (torch-CPU restatement of the reference op sequence), %s, %d shape(s) per step, '
                     '%d warm-up + %d timed steps in %.1f s (+%.1f s warm-up); shape-steps/s / %d'
                     % (w['config'] + ' ' + w['stage'], r1['shapes_per_step'], n_warm, n_timed, r1['dt'], r1['warm'], batch),
           'threads_note': 'threads = min(32, usable CPUs); usable = min(affinity, cgroup CPU quota).  The GPU boxes show '
                           '%d logical CPUs under a quota of %s CPUs: more runnable threads than the quota are throttled '
                           '(256 threads: 140 s per hr step, round 1; 8 pinned 32-thread processes: 32 s per step each, '
                           'round 4), so the whole USABLE box is what one process with that many threads gets'
                           % (ncpu, 'no' if quota is None else '%g' % quota)}
    per = int(os.environ.get('OFX_CPU_WORKER_THREADS', '32'))      # (override: tests on small hosts)
    n_workers = min(8, usable // per)
    if n_workers >= 2:
        nw_warm, nw_timed = max(1, n_warm - 1), max(2, n_timed - 1)
        with tempfile.TemporaryDirectory() as sd:
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--cpu-worker', name, str(batch), str(k),
                                       str(n_workers), str(per), str(nw_warm), str(nw_timed), sd],
                                      stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                      env=dict(os.environ, OMP_NUM_THREADS=str(per), MKL_NUM_THREADS=str(per)))
                     for k in range(n_workers)]
            outs = []
            deadline = time.time() + 150.0             # bounded: the default bench run has to finish within minutes
            for p_ in procs:
                try:
                    o, _ = p_.communicate(timeout=max(1.0, deadline - time.time()))
                    outs.append(json.loads([l for l in o.splitlines() if l.startswith('{')][-1]))
                except Exception:      # noqa: BLE001
                    p_.kill()
        if len(outs) == n_workers:
            # every worker runs its timed steps while all the others do (common start after the warm-ups): the box's
            # rate is the sum of the workers' rates
            rate = sum(o['shapes_per_step'] * nw_timed / o['dt'] for o in outs) / batch
            overlap = min(o['wall_end'] for o in outs) - max(o['wall_start'] for o in outs)
            res['whole_box'] = {'value': rate, 'processes': n_workers, 'threads_per_process': per,
                                'cores': per * n_workers, 'pinned': 'disjoint blocks of %d logical CPUs (sched_setaffinity)' % per,
                                'timed_s_per_worker': [o['dt'] for o in outs], 'steps_per_worker': nw_timed,
                                'common_window_s': overlap,
                                'sample': '%d concurrent workers, one shape each, %d warm-up + %d timed steps' % (n_workers, nw_warm, nw_timed)}
            if rate > res['value']:
                res['value'], res['cores'] = rate, per * n_workers
                res['sample'] += '; value = whole box: %d pinned %d-thread processes, one shape each' % (n_workers, per)
        else:
            res['whole_box'] = {'error': '%d of %d workers returned' % (len(outs), n_workers)}
    return res


def parity_spot_check(wl, dev):
    """One FULL-SIZE layer of the benchmarked net on the benchmark's own tree against the CPU oracle: the first
    res-block's conv1 (GroupNorm + SiLU -> GraphConv 128 -> 128 at the input depth, the planes kernel's path)."""
    from octfusion_amd import synthetic
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    net = getattr(wl.net, 'unet_' + wl.stage)
    blk = net.input_blocks[1]
    d = net.input_depth
    split = synthetic.shell6_split(wl.batch, jitter=True)
    oc = OS.split2octree_small(split, 6, 4)
    if wl.w['tree'] == 'shell8':
        x6, y6, z6, _ = oc.xyzb(6)
        oc = OS.split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), 6)
    o_doc = OD.OracleDualOctree(oc)
    o_doc.post_processing_for_docnn()
    C = blk.channels
    g = torch.Generator().manual_seed(7)
    x = torch.randn(wl.doc.csr(d)[2], C, generator=g)
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    h_ref = OM.silu(OM.dual_octree_group_norm(x, o_doc, d, sd['block1_norm.weights'], sd['block1_norm.bias']))
    y_ref = OM.graph_conv(h_ref, o_doc, d, sd['conv1.weights'], None, blk.conv1.n_node_type)
    h = blk.block1_norm(x.to(dev), wl.doc, d, act='silu', planes=blk.conv1.planes_mode(wl.doc, d))
    y = blk.conv1(h, wl.doc, d).cpu()
    err = float((y - y_ref).abs().max() / y_ref.abs().max())
    return {'layer': 'input_blocks.1: GroupNorm+SiLU -> GraphConv %d -> %d at depth %d, N = %d' % (
        C, blk.out_channels, d, x.shape[0]), 'planes_kernel': bool(getattr(h, '_ofx_planes', 0)),
        'rel_to_max_vs_oracle': err, 'bound': 1e-3}


def mfma_sustained_probe(dev, steps=20000):
    """The matrix-pipe rate this box SUSTAINS with realistic fp16 hi / lo operand pairs (csrc/ofx_probe.hip): the
    data-sheet peak assumes the boost clock, which the chip only holds with constant operands."""
    from octfusion_amd import _lib
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    ticks = torch.zeros(1, dtype=torch.int64, device=dev)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    args = (cus, _lib.ptr(sink), _lib.ptr(ticks), _lib.stream())
    _lib.call('ofx_probe_mfma_sustained', steps // 10, *args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    _lib.call('ofx_probe_mfma_sustained', steps, *args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    tf = cus * 8.0 * steps * 24 * 32768 / (ms * 1e-3) / 1e12        # 32 x 32 x 16 x 2 flops per MFMA
    return {'issued_TFLOPs': tf, 'shader_clock_GHz': float(ticks.item()) / (ms * 1e6), 'ms': ms, 'steps': steps,
            'what': 'v_mfma_f32_32x32x16_f16 x 24 per wave per step on four accumulators, 2 waves per SIMD on every CU, '
                    'operands = fp16 hi / lo pairs of N(0,1)-like activations and of weights scaled into [2^14, 2^15) read '
                    'from LDS (ds_read_b128), one s_barrier per step, NO global memory traffic: the ceiling the three-term '
                    'contraction could reach on this box if everything but the MFMAs were free (csrc/ofx_probe.hip; '
                    'constant operands reach 2.4 PFLOP/s at 2.3 GHz: profiles/r04/mfma_rate_probe.txt)'}


def gather_microbench(doc, dev, C=128, iters=20):
    """Stand-alone segment-mean gather (the reference's col_data) at depth 6: HBM GB/s."""
    from octfusion_amd import ops
    seg_ptr, col, N, E = doc.csr(6)
    x = torch.randn(N, C, device=dev)
    for _ in range(3):
        ops.gather_mean(x, seg_ptr, col)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gather_mean(x, seg_ptr, col)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = 4.0 * E * C + 8.0 * E + 4.0 * N * 7 * C          # SURVEY 8(d) stand-alone gather bytes
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {'kernel': 'gather_mean_kernel', 'C': C, 'N': N, 'E': E, 'ms': ms, 'achieved': gbs, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}


KERNEL_SOURCES = ('ofx_gemm3.hip', 'ofx_planes.h', 'ofx_gemm2.hip', 'ofx_gemm.hip', 'ofx_gemm_common.h')
PMC_FILE = os.path.join(ROOT, 'profiles', 'r06', 'pmc_traffic.json')


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, 'octfusion_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def profile_summary(prof, dt_s, kinds, peak_tf):
    """Aggregate the per-launch HIP-event records of the GraphConv launches of the given kinds."""
    sel = [p_ for p_ in prof if p_[5][0] in kinds]
    if not sel:
        return None
    t_ms = sum(a.elapsed_time(b) for a, b, *_ in sel)
    flops = sum(p_[2] for p_ in sel)
    nbytes = sum(p_[3] for p_ in sel)
    n = len(sel)
    t_s = t_ms * 1e-3
    return {'launches': n, 'avg_launch_us': 1e3 * t_ms / n, 'algorithmic_flops_per_launch': flops / n,
            'algorithmic_bytes_per_launch': nbytes / n, 'algorithmic_TFLOPs': flops / t_s / 1e12,
            'mfma_peak_for_algorithmic_flops_TFLOPs': peak_tf, 'mfma_frac': flops / t_s / 1e12 / peak_tf,
            'algorithmic_GBps': nbytes / t_s / 1e9, 'hbm_frac': nbytes / t_s / 1e9 / HBM_PEAK_GBS,
            'time_frac_of_step': t_s / dt_s}


TAIL_BOUND = {'dense_gemm': 'mfma', 'gridconv_27tap': 'mfma', 'attention': 'mfma'}       # every other class: HBM


def tail_summary(tail, steps, peak_tf, step_ms, eager_ms):
    """roofline_tail: everything of a step that is NOT the fused GraphConv, per kernel class.  `tail` = the
    (entry point, start event, end event, meta) records of _lib.PROFILE over `steps` eager steps; a class's time is
    the sum of its entry-point brackets (a bracket includes the launch's own second-stage kernels: split-K reduce,
    statistics reduce); flops / bytes are the ALGORITHMIC ones of the operator (ops._meta)."""
    cls = {}
    shapes = {}
    conv_ms = 0.0
    for name, e0, e1, meta in tail:
        ms = e0.elapsed_time(e1)
        if name in ('ofx_graphconv_fwd_planes', 'ofx_graphconv_fwd'):
            conv_ms += ms
            continue
        kind, fl, nb, tag = meta if meta is not None else (name.replace('ofx_', ''), 0.0, 0.0, None)
        c = cls.setdefault(kind, [0, 0.0, 0.0, 0.0, 0.0])
        c[0] += 1; c[1] += ms; c[2] += fl; c[3] += nb
        c[4] += 1e3 * max(nb / (HBM_PEAK_GBS * 1e9), fl / (peak_tf * 1e12))       # the launch's own ideal time
        if tag is not None:
            sh = shapes.setdefault((kind,) + tuple(tag), [0, 0.0, fl, nb])
            sh[0] += 1; sh[1] += ms
    rows = {}
    tot_ms = tot_n = 0.0
    for kind, (n, ms, fl, nb, ideal_ms) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
        bound = TAIL_BOUND.get(kind, 'hbm')
        t_s = ms * 1e-3
        r = {'launches_per_step': n / steps, 'ms_per_step': ms / steps, 'bound': bound,
             'algorithmic_GFLOP_per_step': fl / steps / 1e9, 'algorithmic_MB_per_step': nb / steps / 1e6,
             'TFLOPs': fl / t_s / 1e12 if t_s else None, 'GBps': nb / t_s / 1e9 if t_s else None}
        r['frac'] = (r['TFLOPs'] / peak_tf if bound == 'mfma' else r['GBps'] / HBM_PEAK_GBS) if t_s else None
        # the class against max(bytes / HBM peak, flops / matrix peak) taken PER LAUNCH: a class like dense_gemm mixes
        # matrix-bound members (unpool, qkv) with HBM-bound ones (the 1x1 skip convolutions read x once: 3-4 flops per byte)
        r['frac_max_rule'] = ideal_ms / ms if ms else None
        rows[kind] = r
        tot_ms += ms; tot_n += n
    top = sorted(shapes.items(), key=lambda kv: -kv[1][1])[:16]
    return {'classes': rows, 'entry_point_calls_per_step': tot_n / steps, 'ms_per_step': tot_ms / steps,
            'graphconv_entry_ms_per_step': conv_ms / steps,
            # time of an UN-instrumented eager step that no entry-point bracket accounts for (torch-native copies / fills,
            # host gaps between launches); the instrumented pass itself is slower (two event records per call: on the
            # 75-launch lr step that overhead alone was 11.9 ms in round 5 and used to be printed under this name)
            'unattributed_ms_per_step': max(0.0, eager_ms - (tot_ms + conv_ms) / steps),
            'instrumented_pass_ms_per_step': step_ms,
            'peaks': {'mfma_TFLOPs_for_algorithmic_flops': peak_tf, 'hbm_GBps': HBM_PEAK_GBS},
            'top_shapes': [{'op': list(k), 'launches_per_step': v[0] / steps, 'ms_per_step': v[1] / steps,
                            'TFLOPs': v[2] * v[0] / (v[1] * 1e-3) / 1e12 if v[1] and v[2] else None,
                            'GBps': v[3] * v[0] / (v[1] * 1e-3) / 1e9 if v[1] else None} for k, v in top],
            'note': 'HIP events around every libofx entry-point call of the eager re-run (events between calls add '
                    'host gaps: "unattributed" = eager step time minus the brackets = torch-native copies / fills + '
                    'gaps); kernel-symbol view: profiles/r06/bench_r06_<workload>_kernel_stats.csv'}


def per_layer(prof):
    agg = {}
    for a, b, f, nb, _, shp in prof:
        t, c, ff, bb = agg.get(shp, (0.0, 0, 0.0, 0.0))
        agg[shp] = (t + a.elapsed_time(b), c + 1, f, nb)
    rows = []
    for shp, (t, c, f, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        rows.append({'layer': list(shp), 'launches': c, 'avg_us': 1e3 * t / c, 'TFLOPs': f * c / t / 1e9,
                     'GBps': nb * c / t / 1e6, 'total_ms': t})
    return rows


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--cpu-worker':
        return cpu_worker_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='hr', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='shapes per GPU (weak scaling)')
    ap.add_argument('--precision', default='fp16x3', choices=['fp16x3', 'bf16x3', 'fp32', 'fp16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='time eager launches instead of hipGraph replay')
    ap.add_argument('--lanes', type=int, default=None, help='lanes of a graph stage (default: sampler.lane_count; 1 = one stream)')
    ap.add_argument('--lane-cus', type=int, default=None, help='compute units a persistent launch of a lane is planned for')
    ap.add_argument('--no-extras', action='store_true', help='skip the fp32 / old-kernel / graph / sustained side runs')
    ap.add_argument('--layers', action='store_true', help='add the per-layer table of the fused GraphConv launches')
    ap.add_argument('--bootstrap-only', action='store_true',
                    help='initialise the ranks, broadcast a dummy tensor, print the rank table and exit (CPU test)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start one rank per GPU ourselves
        cmd = rank_bootstrap_cmd(sys.argv[1:], args.gpus)
        sys.exit(subprocess.call(cmd))

    # ONE JSON line on stdout, whatever the libraries print: RCCL writes a version banner to the C-level stdout
    # (flushed at exit, i.e. AFTER our line).  File descriptor 1 is pointed at stderr for the life of the process and
    # the result line is written to a private duplicate of the real stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        os.write(real_stdout, (line + '\n').encode())

    from octfusion_amd import dist
    rank, local_rank, world = dist.init('gloo' if args.bootstrap_only and not torch.cuda.is_available() else None)
    if args.bootstrap_only:
        # rendezvous + every collective kind the timed run depends on (broadcast, MAX reduction, all-gather, barrier),
        # the rank table and the shard rule for 64 shapes (train.py:166-185) -- no HIP call; gloo on a CPU-only machine
        t = torch.full((4,), float(rank))
        if torch.cuda.is_available():
            t = t.cuda(local_rank)
        import torch.distributed as td
        table = [(rank, local_rank, str(t.device))]
        if world > 1:
            td.broadcast(t, src=0)
            table = [None] * world
            td.all_gather_object(table, (rank, local_rank, str(t.device)))
        mx = dist.max_over_ranks(float(rank), t.device)
        per_rank = dist.gather_floats(float(len(dist.shard_indices(64, rank, world))), t.device)
        if rank == 0:
            emit(json.dumps({'bootstrap': 'ok', 'world': world, 'max_rank_seen': mx, 'broadcast_value': float(t[0]),
                             'shard_of_10': dist.shard_indices(10, 0, world),
                             'ranks': [list(r_) for r_ in table], 'shapes_of_64_per_rank': per_rank,
                             'shard_of_64_last_rank': dist.shard_indices(64, world - 1, world)}))
        dist.barrier()
        dist.shutdown()
        return
    rccl_note = None
    if world == 1 and torch.cuda.is_available():
        # one GPU: still run the collectives of the multi-GPU path (weight broadcast, timing reductions, barriers) through
        # RCCL, on a one-rank group, so that a single-GPU box executes the same device path the 8-GPU run depends on
        try:
            dist.init(force=True)
            rccl_note = 'one-rank nccl (RCCL) group: weight broadcast, barriers and timing reductions ran through it'
        except Exception as e:      # noqa: BLE001
            rccl_note = 'one-rank RCCL group could not be created (%s): collectives skipped at world 1' % e
    if world != args.gpus and rank == 0:
        print('warning: --gpus %d but WORLD_SIZE %d (reporting n_gpus = %d)' % (args.gpus, world, world), file=sys.stderr)

    from octfusion_amd import _lib, ops
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    _lib.require_device()
    torch.set_grad_enabled(False)
    w = WORKLOADS[args.workload]
    batch = args.batch or w['batch']
    K = args.steps if args.steps is not None else w['steps']
    W = args.warmup
    ops.set_precision(args.precision)

    wl = Workload(args.workload, batch, dev, rank, lanes=args.lanes)
    if wl.lanes:
        ops.set_lane_cus(wl.sampler.LANE_CUS if args.lane_cus is None else args.lane_cus)
    first_ms = 1e3 * timed(lambda: wl.run(0, 1))        # first step of the process: weight packing + per-doctree tables
    new_tree_first_ms = None
    if wl.doc is not None and rank == 0:
        # what every LATER batch of shapes pays on its first step: a new doctree (its lazily built tables and operand
        # slabs) with the weights already packed
        doc_keep, lanes_keep = wl.doc, wl.lanes
        wl.set_doc(build_tree(w['tree'], batch, dev)[1])
        new_tree_first_ms = 1e3 * timed(lambda: wl.run(0, 1))
        wl.doc, wl.lanes = doc_keep, lanes_keep
    wl.run(1, max(W - 1, 0))
    torch.cuda.synchronize()
    steady_ms = 1e3 * timed(lambda: wl.run(W, 1))

    # ---- execution mode of the timed region: the whole step (U-Net forward + DDIM update) captured once into a
    # hipGraph and replayed -- what sampler.sample_loop does by default: every shape is static across the steps of a
    # stage (the doctree is fixed), only x / log-SNR / coefficients / noise change and they live in static buffers.
    # The lr stage has two regimes (with / without the sign() of the truncated steps) -> one graph each, and its
    # self-conditioning input (the previous step's x0 prediction) is copied into a static buffer after every replay.
    # Falls back to eager launches if capture fails.
    replay = None
    if not args.eager:
        try:
            is_lr = wl.stage == 'lr'
            cond_s = wl.cond[0].expand(batch).contiguous().clone()
            coef_s = wl.coef[0].clone()
            noise_s = torch.randn_like(wl.x) if wl.df == 'x0' else None      # multiplied by coef[3] (0 on noise-free steps)
            self_s = torch.zeros_like(wl.x) if is_lr else None               # no self-conditioning on the first step = zeros

            def capture(sign):
                def gstep():
                    return wl.sampler._step(wl.net, wl.x, cond_s, wl.stage, wl.df, wl.doc, wl.nested, wl.label, self_s,
                                            coef_s, noise_s, sign, None)
                gph = torch.cuda.CUDAGraph()
                side_s = ops.side_stream(dev)
                side_s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side_s):
                    gstep()
                    gstep()
                torch.cuda.current_stream().wait_stream(side_s)
                with torch.cuda.graph(gph, stream=side_s):      # the warm-up stream: its scratch exists already (ops._no_capture)
                    out = gstep()
                return gph, out
            if wl.lanes:
                capture_lanes(wl)
            else:
                graphs = {sign: capture(sign) for sign in sorted(set(wl.sign if is_lr else [False]))}

            def replay(first, n):
                if wl.lanes:
                    return replay_lanes(wl, first, n)
                for i in range(first, first + n):
                    if i % wl.RESET_EVERY == 0 and i > 0:
                        wl.x.copy_(wl.x_init)
                        if self_s is not None:
                            self_s.zero_()
                    j = i % 200
                    cond_s.copy_(wl.cond[j].expand(batch))
                    coef_s.copy_(wl.coef[j])
                    if noise_s is not None:
                        noise_s.normal_()
                    gph, out = graphs[wl.sign[j] if is_lr else False]
                    gph.replay()
                    if is_lr:
                        self_s.copy_(out)
            replay(0, 2)
            torch.cuda.synchronize()
        except Exception as e:      # noqa: BLE001
            print('hipGraph capture failed (%s): timing eager launches' % e, file=sys.stderr)
            replay = None
    run_timed = replay if replay is not None else wl.run

    # ---- the contract region: exactly K steps, barrier + synchronize on both sides, MAX over ranks --------
    prof = []
    if replay is None:
        ops.GRAPHCONV_PROFILE = prof
    dist.barrier()
    torch.cuda.synchronize()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    run_timed(W + 1, K)
    cpu_issue = time.process_time() - cpu0          # host CPU time to ISSUE the K steps (before waiting for the GPU)
    torch.cuda.synchronize()
    dist.barrier()
    dt_local = time.perf_counter() - t0
    cpu_total = time.process_time() - cpu0
    ops.GRAPHCONV_PROFILE = None
    dt = dist.max_over_ranks(dt_local, dev)
    rank_ms = dist.gather_floats(1e3 * dt_local / K, dev)
    ms_step = 1e3 * dt / K
    eager_ms = None
    dt_prof = dt
    if replay is not None:
        # HIP events cannot be recorded inside a graph replay: the per-launch timings of the roofline come from an
        # eager re-run of the same K steps right after the timed region (same kernels, same inputs)
        # (lanes: one after another in this pass, so that an event pair brackets a kernel that has the chip to itself)
        wl.serial = True
        ops.GRAPHCONV_PROFILE = prof
        dt_prof = timed(lambda: wl.run(W + 1, K))
        ops.GRAPHCONV_PROFILE = None
        wl.serial = False
        eager_ms = 1e3 * dt_prof / K
    # per-class accounting of everything that is not the fused GraphConv: a second eager pass with an event pair
    # around every entry-point call (kept apart from the pass above so that its brackets stay undisturbed)
    tail, n_tail = [], max(2, min(K, 5))
    if rank == 0 and not args.no_extras:
        wl.serial = True
        _lib.PROFILE = tail
        dt_tail = timed(lambda: wl.run(W + 1, n_tail))
        _lib.PROFILE = None
        wl.serial = False

    assert wl.finite()
    ops.raise_on_sync_error(dev)            # a flag wait of the persistent launch gave up: the timing would be void

    res = None
    if rank == 0:
        bf = args.precision
        peak = PEAKS[bf]
        planes_kind = {'fp16x3': 'graph2', 'bf16x3': 'graph2', 'fp16': 'graph2h'}.get(bf)
        dom = profile_summary(prof, dt_prof, (planes_kind,), peak) if planes_kind else None
        dom_is_planes = dom is not None
        dom_name = ('gconv3_kernel<%d,*,*> (fused GraphConv on operand planes, persistent stream-K blocks: LDS-DMA gather '
                    '-> %s MFMA, fp32 accumulate; gconv2_kernel for the layers too small for it)'
                    % {'fp16x3': (3, 'fp16x3 (three fp16 MFMAs per product)'), 'bf16x3': (2, 'bf16x3'), 'fp16': (1, 'fp16')}[bf])
        if dom is None:         # exact-fp32 mode / dense lr stage: the register-staged kernel carries the time
            dom = profile_summary(prof, dt_prof, ('graph', 'grid'), peak)
            dom_name = 'gemm_fast_kernel / gemm_pairs_x3_kernel<MODE_GATHER> (register-staged fused GraphConv / 27-tap gridconv)'
        graph_only = profile_summary(prof, dt_prof, ('graph', 'graph2', 'graph2h'), peak)
        grid_only = profile_summary(prof, dt_prof, ('grid',), peak)
        roof = {'kernel': dom_name}
        if dom:
            # which resource binds: measured HBM traffic of the kernel is well below its algorithmic bytes (L2 absorbs
            # the 7x neighbour re-reads, profiles/), so the bf16x3 / fp32 kernels are judged against the MATRIX roof;
            # the single-pass fp16 kernel (3x fewer MFMAs per byte) against HBM.
            # the roof is max(bytes / HBM peak, flops / matrix peak) -- at one measured time the binding
            # resource is the one with the LARGER fraction.  hr: matrix pipe; the depth-8 feature stage (64-wide layers:
            # 7 x re-gathered rows per few flops) and the single-pass fp16 mode: HBM.
            # The kernel is priced against the matrix roof unless HBM clearly binds (hbm_frac > 1.25 x mfma_frac; on the hr
            # workload the two ideal times are within 7 % of each other and the line keeps the yardstick of rounds 1-5).
            bound = 'hbm' if dom['hbm_frac'] > 1.25 * dom['mfma_frac'] else 'mfma'
            roof.update({'bound': bound, 'bound_rule': 'SURVEY 8d max(algorithmic bytes / 8 TB/s, algorithmic flops / matrix '
                                                       'peak): mfma_frac %.3f, hbm_frac %.3f -> %s (hbm only if > 1.25 x mfma)'
                                                       % (dom['mfma_frac'], dom['hbm_frac'], bound),
                         'achieved': dom['algorithmic_GBps'] if bound == 'hbm' else dom['algorithmic_TFLOPs'],
                         'peak': HBM_PEAK_GBS if bound == 'hbm' else peak,
                         'unit': 'GB/s' if bound == 'hbm' else 'TFLOP/s',
                         'frac': dom['hbm_frac'] if bound == 'hbm' else dom['mfma_frac'],
                         'traffic': None})
            roof.update(dom)
            # counters of the workload's OWN shapes (hr: shell-6 x 8, hr_cond: shell-6 x 4, feature: shell-8 x 8)
            tpath = PMC_FILE.replace('.json', '_%s.json' % args.workload) if args.workload in ('feature', 'hr_cond') else PMC_FILE
            if dom_is_planes and os.path.exists(tpath):       # (the counters are the planes GraphConv's: nothing to say about the dense lr stage)
                try:
                    pj = json.load(open(tpath))
                    want_k = {'fp16x3': 'gconv3_kernel<3,', 'bf16x3': 'gconv3_kernel<2,', 'fp16': 'gconv3_kernel<1,'}.get(bf)
                    kernels = [L_.get('kernel', '') for L_ in pj.get('layers', [])]
                    if not kernels or not all(want_k and want_k in k_ for k_ in kernels):
                        # counters taken on another instantiation than the one this run timed say nothing about it
                        roof['traffic_note'] = ('profiles/r06/pmc_traffic*.json holds counters of %s, this run timed %s...>: '
                                                'not reported' % (sorted(set(kernels)), want_k))
                    elif pj.get('kernel_source_sha16') == kernel_source_hash():
                        # counters exist for four probe layers (tools/pmc_probe2.py); `traffic` is the HBM byte count
                        # of the depth-6 128 -> 128 layer, next to that layer's own algorithmic bytes
                        roof['traffic'] = pj.get('hbm_bytes_per_launch')
                        roof['traffic_layer'] = pj.get('hbm_bytes_per_launch_layer')
                        L0 = pj['layers'][0]
                        if args.workload == 'feature':       # depth-8 64 -> 64 layer of the shell-8 x 8 tree
                            n8, e8 = wl.doc.csr(8)[2], wl.doc.csr(8)[3]
                            roof['traffic_layer_algorithmic_bytes'] = 4.0 * (e8 * 64 + n8 * 64 + 7 * 71 * 64) + 8.0 * e8
                        else:                                # depth-6 128 -> 128 layer of this workload's shell-6 batch
                            n6, e6 = wl.doc.csr(6)[2], wl.doc.csr(6)[3]
                            roof['traffic_layer_algorithmic_bytes'] = 4.0 * (e6 * 128 + n6 * 128 + 931 * 128) + 8.0 * e6
                        roof['traffic_per_layer'] = [{'layer': L_['layer'], 'kernel': L_['kernel'], 'hbm_bytes': L_['hbm_bytes_per_launch'],
                                                      'mfma_busy_of_clocked_cycles': L_['mfma_busy_frac_of_clocked_simd_cycles'],
                                                      'clock_ghz': L_['gpu_clock_ghz_under_kernel']} for L_ in pj['layers']]
                        roof['traffic_source'] = pj.get('source')
                        roof['mfma_pmc'] = pj.get('mfma')
                    else:
                        roof['traffic_note'] = ('profiles/r06/pmc_traffic*.json was measured on kernel sources %s, this '
                                                'build is %s: not reported' % (pj.get('kernel_source_sha16'), kernel_source_hash()))
                except Exception as e:      # noqa: BLE001
                    roof['traffic_note'] = 'pmc_traffic.json unreadable: %s' % e
            if wl.lanes:
                # what the overlap is worth, in the kernel's own unit: the GraphConv flops of a step over the WHOLE step time
                fl = dom['algorithmic_flops_per_launch'] * dom['launches'] / K / 1e9          # GFLOP per step (TFLOP/s x ms)
                roof['whole_step'] = {'graphconv_GFLOP_per_step': fl, 'TFLOPs_over_the_timed_step': fl / ms_step,
                                      'frac_of_matrix_peak': fl / ms_step / peak,
                                      'same_over_the_serial_eager_step': fl / eager_ms / peak if eager_ms else None}
            roof['note'] = ('per launch, HIP events on the launching stream '
                            + ('in an eager re-run of the same K steps right after the hipGraph-replayed timed region '
                               if replay is not None else 'inside the timed region ')
                            + '(bracket includes the fused-statistics second-stage reduce and, for inputs not produced '
                            'by a GroupNorm, the multi-neighbour pre-pass)')
        if tail:
            res_tail = tail_summary(tail, n_tail, peak, 1e3 * dt_tail / n_tail, eager_ms if eager_ms is not None else ms_step)
        # the network's input / output convolutions (3 or 8 channels on one side) are gathers with almost no arithmetic:
        # they are judged against HBM on their own line, not inside the matrix-bound aggregate
        narrow = [r_ for r_ in per_layer(prof) if r_['layer'][0].startswith('graph') and min(r_['layer'][2], r_['layer'][3]) <= 8]
        roof['narrow_graphconv_hbm'] = [{'layer': r_['layer'], 'launches': r_['launches'], 'avg_us': r_['avg_us'],
                                         'algorithmic_GBps': r_['GBps'], 'peak': HBM_PEAK_GBS, 'frac': r_['GBps'] / HBM_PEAK_GBS,
                                         'bound': 'hbm',
                                         'note': 'input convolution: the operator\'s bytes (one cin-wide source row per edge + the output); '
                                                 'output convolution: the bytes of project-then-aggregate as it runs -- x, P written and '
                                                 'read once, the CSR, the output (ops.graphconv_narrow_out)'} for r_ in narrow]
        roof['all_graphconv_launches'] = graph_only
        roof['gridconv_27tap_launches'] = grid_only
        res = {
            'metric': 'denoising-steps/sec (depth-8 octree, batch 8)', 'value': world * K / dt,
            'unit': 'steps/s', 'n_gpus': world, 'steps': K, 'warmup': W, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp16x3': 'f32 storage, fp16x3 products (fp16 hi + lo operand pairs = 22 significand bits, three fp16 '
                                'MFMAs per product, fp32 accumulate)',
                      'bf16x3': 'f32 storage, bf16x3 products (16-bit significand pairs, fp32 accumulate)',
                      'fp32': 'f32', 'fp16': 'f32 storage, fp16 products in GraphConv (reduced precision)'}[bf],
            'contraction': bf, 'data': 'synthetic',
            'execution': ('hipGraph replay of the captured step' if replay is not None else 'eager launches')
                         + (': %d lanes (runs of %s shapes) on %d HIP streams, one hipGraph per lane (sampler.sample_loop\'s default for '
                            'graph stages)' % (len(wl.lanes), '/'.join(str(L.batch) for L in wl.lanes), len(wl.lanes)) if wl.lanes else ''),
            'lanes': {'n': len(wl.lanes), 'shapes': [L.batch for L in wl.lanes], 'rows': [int(L.x.shape[0]) for L in wl.lanes],
                      'persistent_launch_planned_for_cus': (wl.sampler.LANE_CUS if args.lane_cus is None else args.lane_cus) or 'all',
                      'split_ms_per_batch_of_shapes': wl.lane_split_ms,
                      'note': 'the shapes of a batch share nothing in the net, so a stage runs as independent half-batches whose '
                              'matrix-bound and HBM-bound launches overlap; `roofline` and `roofline_tail` are per-launch event '
                              'timings of an eager pass with the lanes one after another (a kernel alone on the chip, at the '
                              'lane\'s row count); --lanes 1 runs the whole batch on one stream'} if wl.lanes else None,
            'eager_ms_per_step': eager_ms if replay is not None else ms_step,
            'config': {'workload': w['desc'] % batch, 'name': args.workload, 'config': w['config'],
                       'batch_per_gpu': batch, 'nodes_per_gpu': wl.doc.total_num if wl.doc else None,
                       'parallelism': 'batch-shard x%d, one RCCL weight broadcast (%d bytes)' % (world, wl.bcast_bytes)},
            'shape_steps_per_s': world * batch * K / dt,
            'per_rank_ms_per_step': rank_ms,
            # what a rank costs the HOST (8 ranks share the node's CPU quota: 16 CPUs on this pool): process CPU time per
            # step to issue the work, and including the wait for the GPU (torch's synchronize spins)
            'host_cpu_ms_per_step': {'issue': 1e3 * cpu_issue / K, 'issue_plus_sync_wait': 1e3 * cpu_total / K},
            'weight_broadcast_bytes': wl.bcast_bytes,
            'rccl': rccl_note if world == 1 else 'nccl (RCCL) group of %d ranks' % world,
            'per_shape_setup': {'octree_and_dual_graph_ms': wl.setup_warm_ms, 'octree_and_dual_graph_first_call_ms': wl.setup_ms,
                                'first_step_ms': first_ms, 'first_step_of_a_later_batch_ms': new_tree_first_ms,
                                'steady_step_ms': steady_ms,
                                'note': 'once per batch of shapes: octree + dual-graph build (host-synchronised) and a first step '
                                        'that builds the per-doctree gather tables; the very first step of a process also packs '
                                        'the weights'},
            'roofline': roof,
        }
        if tail:
            res['roofline_tail'] = res_tail
        if args.layers:
            res['layers'] = per_layer(prof)

    # ---- side measurements (single GPU only; after the contract region) -------------------------------------
    if world == 1 and rank == 0 and not args.no_extras:
        # (before the side runs: they re-pack weights, and the captured graph keeps raw pointers to the current packs)
        # sustained run: keep the GPU busy for >= 3 s so an outside observer (rocm-smi samples) sees the load
        n_sus = max(K, int(3000.0 / max(ms_step, 0.05)) + 1)
        t = timed(lambda: run_timed(0, n_sus))
        res['sustained'] = {'steps': n_sus, 'seconds': t, 'ms_per_step': 1e3 * t / n_sus,
                            'mode': 'hipGraph replay' if replay is not None else 'eager'}
        res['hipgraph_replay_ms_per_step'] = ms_step if replay is not None else None
        n_side = max(5, min(K, 20))
        if wl.doc is not None and not wl.lanes and batch >= 2 and replay is not None:
            # (before the side runs, like the sustained run: they re-pack the weights the captured graphs point at)
            # in-run A/B of sampler.sample_loop's opt-in lanes (OFX_LANES=2): the same step as two half-batches on two HIP
            # streams, one hipGraph each, the persistent launches planned for LANE_CUS compute units
            try:
                base = 1e3 * timed(lambda: run_timed(0, n_side)) / n_side
                wl.n_lanes = 2
                ops.set_lane_cus(wl.sampler.LANE_CUS)
                wl.make_lanes()
                wl.run(0, 2)
                capture_lanes(wl)
                replay_lanes(wl, 0, 2)
                two = 1e3 * timed(lambda: replay_lanes(wl, 2, n_side)) / n_side
                res['lanes_ab'] = {'one_lane_ms_per_step': base, 'two_lanes_ms_per_step': two, 'steps': n_side,
                                   'planned_for_cus': wl.sampler.LANE_CUS, 'finite': wl.finite(),
                                   'note': 'hipGraph replay both; off by default (octfusion_amd/sampler.py: LANES)'}
            except Exception as e:      # noqa: BLE001
                res['lanes_ab'] = {'error': str(e)}
            finally:
                ops.set_lane_cus(0)
                wl.lanes, wl.n_lanes = [], 1
        extras = {}

        def side(label, precision, planes, persistent=1):
            ops.set_precision(precision)
            ops.USE_PLANES = planes
            _lib.call('ofx_set_gconv_persistent', persistent)
            wl.run(0, 2)
            p2 = []
            ops.GRAPHCONV_PROFILE = p2
            t = timed(lambda: wl.run(2, n_side))
            ops.GRAPHCONV_PROFILE = None
            pk = PEAKS[precision]
            s = profile_summary(p2, t, ('graph', 'graph2', 'graph2h'), pk)
            extras[label] = {'ms_per_step': 1e3 * t / n_side, 'steps': n_side,
                             'graphconv_TFLOPs': s and s['algorithmic_TFLOPs'], 'graphconv_mfma_frac': s and s['mfma_frac'],
                             'graphconv_GBps': s and s['algorithmic_GBps'], 'graphconv_hbm_frac': s and s['hbm_frac']}
        try:
            side('fp32_exact', 'fp32', False)
            res['fp32_ms_per_step'] = extras['fp32_exact']['ms_per_step']
            res['fp32_mfma_frac'] = extras['fp32_exact']['graphconv_mfma_frac']
            # the same-arithmetic-as-the-reference figure as a peer of `value` (eager launches; `value` is graph replay)
            res['value_fp32_exact'] = world * 1e3 / extras['fp32_exact']['ms_per_step']
            res['roofline_fp32_exact'] = {'bound': 'mfma', 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                          'achieved': extras['fp32_exact']['graphconv_TFLOPs'],
                                          'frac': extras['fp32_exact']['graphconv_mfma_frac'],
                                          'kernel': 'gemm_fast_kernel<MODE_GATHER> (exact fp32 MFMA, register-staged)'}
            if args.precision == 'fp16x3' and wl.doc is not None:
                # in-run A/B of the launch shape (boxes differ by several per cent: only same-run pairs compare):
                # same kernels' data path, one tile per block (csrc/ofx_gemm2.hip) instead of persistent stream-K blocks
                side('fp16x3_eager', 'fp16x3', True)
                side('fp16x3_one_tile_per_block_launch', 'fp16x3', True, persistent=0)
                side('fp16x3_pure_stream_k_launch', 'fp16x3', True, persistent=2)
                side('bf16x3', 'bf16x3', True)
                side('fp16x3_register_staged_kernel', 'fp16x3', False)
                side('fp16_single_pass', 'fp16', True)
        finally:
            ops.set_precision(args.precision)
            ops.USE_PLANES = True
            _lib.call('ofx_set_gconv_persistent', 1)
        res['side_runs'] = extras
        if 'fp16_single_pass' in extras:
            # BASELINE configs[4] names "fp16 MFMA": the single-pass fp16 contraction as a peer of `value` (eager launches).
            # It is NOT the shipped default: element-wise p99.9 2e-2 against the fp64 oracle (whole-step rel-to-max 4.5e-4,
            # inside north_star's 1e-3) -- DESIGN section 2, tests/test_gpu_cascade.py::test_cascade_in_fp16_single_pass
            fx = extras['fp16_single_pass']
            res['value_fp16_single_pass'] = world * 1e3 / fx['ms_per_step']
            res['roofline_fp16_single_pass'] = {'bound': 'hbm', 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'achieved': fx['graphconv_GBps'],
                                                'frac': fx['graphconv_hbm_frac'], 'mfma_frac_of_2500': fx['graphconv_mfma_frac'],
                                                'kernel': 'gconv3_kernel<1, *, *> (one v_mfma_f32_32x32x16_f16 per product)'}
        try:
            sus = mfma_sustained_probe(dev)
            roof = res['roofline']
            roof['sustained'] = sus
            if roof.get('bound') == 'mfma' and roof.get('achieved') and args.precision != 'fp32':
                n_mfma = 3.0 if args.precision in ('fp16x3', 'bf16x3') else 1.0
                roof['sustained']['roof_for_algorithmic_flops_TFLOPs'] = sus['issued_TFLOPs'] / n_mfma
                roof['frac_of_sustained'] = roof['achieved'] * n_mfma / sus['issued_TFLOPs']
        except Exception as e:      # noqa: BLE001
            res['roofline']['sustained'] = {'error': str(e)}

        if wl.doc is not None and 6 in wl.doc._csr:
            res['gather'] = gather_microbench(wl.doc, dev)
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        if wl.doc is not None:
            res['parity_spot_check'] = parity_spot_check(wl, dev)
            assert res['parity_spot_check']['rel_to_max_vs_oracle'] < 1e-3, res['parity_spot_check']
        res['cpu_baseline'] = cpu_baseline(args.workload, batch)
        res['gpu_over_cpu'] = res['value'] / res['cpu_baseline']['value']
    if rank == 0:
        emit(json.dumps(res))
    dist.barrier()
    dist.shutdown()


if __name__ == '__main__':
    main()
