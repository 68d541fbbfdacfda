"""GPU parity at the REAL network widths (configs.py = the reference's YAMLs), not the shrunken fixtures:
one denoising step of every BASELINE workload against the CPU oracle, in every contraction mode, plus a sweep
over every (K, Cout) the three configs put through GraphConv.

Error figures (both are printed, the first is asserted against north_star's 1e-3):
  * rel-to-max  = max |a - b| / max |b|                      (what tests/test_gpu_parity.close() uses)
  * element-wise = |a - b| / max(|b|, 1e-2 * max |b|), its 99.9th percentile and maximum -- a per-element
    relative error with a floor at 1 % of the tensor's range, so exact zeros do not divide.
The fp16 single-pass mode (ofx_set_precision(2), BASELINE configs[4] "fp16 MFMA") is reduced precision by
construction: its bound is measured here and asserted at 1e-2 / reported.
"""
import json
import os
import time

import pytest
import torch

import functools

import common as C
import oracle_cache as OC
from oracle_cache import errors          # noqa: F401  (tensor or Sketch reference; re-exported to the other GPU test modules)

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# scratch copy of the error figures of a run (gpurun merges gpurun_out/ back); `tools/collect_profiles.sh` copies it to
# profiles/rNN/fullwidth_parity.jsonl, the tracked evidence file
REPORT = os.path.join(ROOT, 'gpurun_out', 'fullwidth_parity.jsonl')


def dev():
    return torch.device('cuda:0')


def report(rec):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, 'a') as f:
            f.write(json.dumps(rec) + '\n')
    except OSError:
        pass
    print(json.dumps(rec))


# (precision, use the LDS-DMA planes kernel, asserted rel-to-max bound)
MODES = [('fp16x3', True, 1e-3), ('fp16x3', False, 1e-3), ('bf16x3', True, 1e-3), ('fp32', False, 1e-3), ('fp16', True, 1e-2)]


class _Modes:
    def __init__(self, prec, planes):
        self.prec, self.planes = prec, planes

    def __enter__(self):
        from octfusion_amd import ops
        self.saved = (ops.get_precision(), ops.USE_PLANES)
        ops.set_precision(self.prec)
        ops.USE_PLANES = self.planes

    def __exit__(self, *exc):
        from octfusion_amd import ops
        ops.set_precision(self.saved[0])
        ops.USE_PLANES = self.saved[1]


_ORACLE = {}


@functools.lru_cache(maxsize=4)
def shell6_oracle(B, jitter=True):
    """(oracle octree, oracle dual octree) of the synthetic shell-6 batch -- CPU only (what the oracle cases run on)."""
    from octfusion_amd import synthetic
    from oracle import dual_octree as OD, sampler as OS
    o_oc = OS.split2octree_small(synthetic.shell6_split(B, jitter=jitter), 6, 4)
    o_doc = OD.OracleDualOctree(o_oc)
    o_doc.post_processing_for_docnn()
    return o_oc, o_doc


@functools.lru_cache(maxsize=2)
def shell8_oracle():
    """(split_large codes, oracle dual octree) of the synthetic shell-8 tree of ONE shape (N8 = 448 232) -- CPU only."""
    from octfusion_amd import synthetic
    from oracle import dual_octree as OD, sampler as OS
    o6 = OS.split2octree_small(synthetic.shell6_split(1, jitter=False), 6, 4)
    x6, y6, z6, _ = o6.xyzb(6)
    sl = synthetic.shell8_split_large(x6, y6, z6)
    o_doc = OD.OracleDualOctree(OS.split2octree_large(o6, sl, 6))
    o_doc.post_processing_for_docnn()
    return sl, o_doc


def shell6_gpu(B, jitter=True):
    from octfusion_amd import synthetic
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    oc = split2octree_small(synthetic.shell6_split(B, jitter=jitter).to(dev()), 6, 4)
    return oc, DualOctree(oc)


def shell8_gpu():
    from octfusion_amd import synthetic
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_large, split2octree_small
    oc6 = split2octree_small(synthetic.shell6_split(1, jitter=False).to(dev()), 6, 4)
    x6, y6, z6, _ = oc6.xyzb(6)
    sl = synthetic.shell8_split_large(x6.cpu(), y6.cpu(), z6.cpu())
    return DualOctree(split2octree_large(oc6, sl.to(dev()), 6))


def shell6(B):
    """GPU and oracle trees together (tests whose oracle side is cheap enough to run on the GPU box)."""
    oc, doc = shell6_gpu(B)
    o_oc, o_doc = shell6_oracle(B)
    return oc, doc, o_oc, o_doc


def _net_sd(config, stage):
    """Seeded synthetic weights of a config's union net, without a device (CPU; the oracle cases' weights)."""
    from octfusion_amd import configs, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    with torch.device('meta'):
        net = UNet3DModel(**configs.unet_params(config, stage))
    return {k: synthetic.fill_param(k, v.shape) for k, v in net.state_dict().items()}


def _hr_case(config, B, xname, t, channels=3, f64=False):
    """Oracle result of one hr step (+ nested lr) on the jittered shell-6 batch: {'ref'[, 'ref64', 'floor']}."""
    from octfusion_amd import configs
    from oracle import modules as OM, sampler as OS, unet as OU
    _, o_doc = shell6_oracle(B)
    sd = _net_sd(config, 'hr')
    st = configs.stage_cfgs(config)
    x = C.rand_input(xname, o_doc.total_num, channels)
    log_snr = OS.beta_linear_log_snr(torch.full((B,), t))
    label = (torch.arange(B) % 5) if st['hr'].get('num_classes') else None
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr')}
    t0 = time.time()
    out = {'ref': OU.hr_forward(parts['unet_hr'], st['hr'], x, o_doc, log_snr, label, parts['unet_lr'], st['lr'])}
    out['oracle_s'] = time.time() - t0
    if f64:
        p64 = _dbl(parts)
        with OM.working_float(torch.float64):
            out['ref64'] = OU.hr_forward(p64['unet_hr'], st['hr'], x.double(), o_doc, log_snr.double(), label, p64['unet_lr'], st['lr'])
        out['floor'] = errors(out['ref'], out['ref64'])
    return out


def _lr_case(config, B, xname, t, labelled=False, f64=False):
    from octfusion_amd import configs
    from oracle import modules as OM, sampler as OS, unet as OU
    sd = _net_sd(config, 'lr')
    st = configs.stage_cfgs(config)
    x = C.rand_input(xname, B, 8, 16, 16, 16)
    xsc = C.rand_input(xname + '_sc', B, 8, 16, 16, 16)
    ls = OS.beta_linear_log_snr(torch.full((B,), t))
    label = (torch.arange(B) % 5) if labelled else None
    p = OM._sub(sd, 'unet_lr')
    out = {'ref': OU.lr_forward(p, st['lr'], x, ls, xsc, label)}
    if f64:
        p64 = _dbl({'p': p})['p']
        with OM.working_float(torch.float64):
            out['ref64'] = OU.lr_forward(p64, st['lr'], x.double(), ls.double(), xsc.double(), label)
        out['floor'] = errors(out['ref'], out['ref64'])
    return out


def _feature_case(f64=False):
    from octfusion_amd import configs
    from oracle import modules as OM, sampler as OS, unet as OU
    _, o_doc = shell8_oracle()
    sd = _net_sd('obja_uncond', 'feature')
    st = configs.stage_cfgs('obja_uncond')
    x = C.rand_input('fw_feature', o_doc.total_num, 3)
    log_snr = OS.beta_linear_log_snr(torch.full((1,), 0.45))
    parts = {p: OM._sub(sd, p) for p in ('unet_feature', 'unet_hr')}
    t0 = time.time()
    out = {'ref': OU.hr_forward(parts['unet_feature'], st['feature'], x, o_doc, log_snr, None, parts['unet_hr'], st['hr'])}
    out['oracle_s'] = time.time() - t0
    if f64:
        p64 = _dbl(parts)
        with OM.working_float(torch.float64):
            out['ref64'] = OU.hr_forward(p64['unet_feature'], st['feature'], x.double(), o_doc, log_snr.double(), None,
                                         p64['unet_hr'], st['hr'])
        out['floor'] = errors(out['ref'], out['ref64'])
    return out


def _dbl(parts):
    return {k: {kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in v.items()} for k, v in parts.items()}


# ---- the oracle cases of this module (computed by tests/golden/make_oracle_cache.py, CPU only) ---------------------
OC.register('hr_step_snet_uncond', lambda: _hr_case('snet_uncond', 2, 'fw_snet_uncond', 0.6, f64=True))
OC.register('hr_step_snet_cond', lambda: _hr_case('snet_cond', 2, 'fw_snet_cond', 0.6))
def _hr_b8_case():
    """The bench's own size, stored VERBATIM (217 008 x 3 floats = 2.6 MB): one full-tensor comparison per suite run."""
    out = _hr_case('snet_uncond', 8, 'fw_b8', 0.6)
    out['ref'] = OC.Full(out['ref'])
    return out


OC.register('hr_step_b8', _hr_b8_case)
OC.register('obja_hr_step', lambda: _hr_case('obja_uncond', 2, 'fw_obja_hr', 0.35, channels=8))
OC.register('lr_step', lambda: _lr_case('snet_uncond', 4, 'fw_lr', 0.3, f64=True))
OC.register('cond_lr_step', lambda: _lr_case('snet_cond', 4, 'fw_cond_lr', 0.8, labelled=True))
OC.register('feature_step', lambda: _feature_case(f64=True))


@pytest.mark.parametrize('config', ['snet_uncond', 'snet_cond'])
def test_full_width_hr_step(config):
    """configs[2] / configs[3]: stage hr (+ the nested dense lr net) at the real widths, shell-6 B = 2
    (two different shapes), labels b mod 5 for the conditional net.  graph_unet_hr.py:214-281."""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    B = 2
    oc, doc = shell6_gpu(B)
    net = UNet3DModel(**configs.unet_params(config, 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    st = configs.stage_cfgs(config)
    x = C.rand_input('fw_' + config, doc.total_num, 3)
    log_snr = sampler.beta_linear_log_snr(torch.full((B,), 0.6))
    label = (torch.arange(B) % 5) if st['hr'].get('num_classes') else None
    case = OC.get('hr_step_' + config)
    ref, t_or = case['ref'], case['oracle_s']
    for prec, planes, bound in MODES:
        with _Modes(prec, planes):
            y = net(unet_type='hr', x=x.to(dev()), doctree=doc, unet_lr=net.unet_lr, timesteps=log_snr.to(dev()),
                    x_self_cond=None, label=label.to(dev()) if label is not None else None)
        e = errors(y, ref)
        report(dict(test='hr_step', config=config, B=B, N=doc.total_num, precision=prec, planes_kernel=planes,
                    oracle_s=t_or, **e))
        assert e['rel_to_max'] < bound, (prec, planes, e)


def test_full_width_lr_step():
    """configs[1]: stage lr (dense 16^3 net with attention) at the real width, batch 4.  graph_unet_lr.py:184-230."""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    B = 4
    net = UNet3DModel(**configs.unet_params('snet_uncond', 'lr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    x = C.rand_input('fw_lr', B, 8, 16, 16, 16)
    xsc = C.rand_input('fw_lr_sc', B, 8, 16, 16, 16)
    log_snr = sampler.beta_linear_log_snr(torch.full((B,), 0.3))
    ref = OC.get('lr_step')['ref']
    for prec, planes, bound in (MODES[0], MODES[2], MODES[3]):
        with _Modes(prec, planes):
            y = net(unet_type='lr', x=x.to(dev()), timesteps=log_snr.to(dev()), x_self_cond=xsc.to(dev()))
        e = errors(y, ref)
        report(dict(test='lr_step', config='snet_uncond', B=B, precision=prec, planes_kernel=planes, **e))
        assert e['rel_to_max'] < bound, (prec, planes, e)


def test_full_width_hr_step_batch8():
    """The bench's own workload size: configs[2], one whole hr step (+ nested lr) on the jittered shell-6 batch of
    EIGHT shapes (N = 217 008) against the oracle, default precision mode (graph_unet_hr.py:214-281)."""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    B = 8
    oc, doc = shell6_gpu(B)
    assert doc.total_num == 217008
    net = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    x = C.rand_input('fw_b8', doc.total_num, 3)
    log_snr = sampler.beta_linear_log_snr(torch.full((B,), 0.6))
    case = OC.get('hr_step_b8')
    ref, t_or = case['ref'], case['oracle_s']
    y = net(unet_type='hr', x=x.to(dev()), doctree=doc, unet_lr=net.unet_lr, timesteps=log_snr.to(dev()),
            x_self_cond=None, label=None)
    assert torch.is_tensor(ref) and tuple(ref.shape) == (217008, 3)          # the whole tensor, not a sketch
    e = errors(y, ref)
    report(dict(test='hr_step_b8', config='snet_uncond', B=B, N=doc.total_num, precision='default', oracle_s=t_or,
                compared_rows=int(ref.shape[0]), **e))
    assert e['rel_to_max'] < 1e-3, e
    from octfusion_amd import ops
    assert not ops.sync_error(dev())


def test_full_width_obja_hr_and_cond_lr_steps():
    """The two stand-alone stages the nested runs do not cover at full width: the Objaverse hr stage (8-channel split
    codes in and out, x0 prediction, num_res_blocks [2, 2, 0]; configs/octfusion_obja_uncond.yaml:11-19) with its lr net
    nested, and the conditional ShapeNet lr stage (channel_mult [1, 2, 4, 8]: a 2^3 level with attention, 5 classes;
    configs/octfusion_snet_cond.yaml:19-25)."""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    B = 2
    oc, doc = shell6_gpu(B)
    net = UNet3DModel(**configs.unet_params('obja_uncond', 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    x = C.rand_input('fw_obja_hr', doc.total_num, 8)
    log_snr = sampler.beta_linear_log_snr(torch.full((B,), 0.35))
    ref = OC.get('obja_hr_step')['ref']
    assert tuple(ref.shape) == (doc.total_num, 8)
    y = net(unet_type='hr', x=x.to(dev()), doctree=doc, unet_lr=net.unet_lr, timesteps=log_snr.to(dev()),
            x_self_cond=None, label=None)
    e = errors(y, ref)
    report(dict(test='obja_hr_step', B=B, N=doc.total_num, precision='default', **e))
    assert e['rel_to_max'] < 1e-3, e

    Bl = 4
    net = UNet3DModel(**configs.unet_params('snet_cond', 'lr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    xl = C.rand_input('fw_cond_lr', Bl, 8, 16, 16, 16)
    xsc = C.rand_input('fw_cond_lr_sc', Bl, 8, 16, 16, 16)
    ls = sampler.beta_linear_log_snr(torch.full((Bl,), 0.8))
    label = torch.arange(Bl) % 5
    ref = OC.get('cond_lr_step')['ref']
    y = net(unet_type='lr', x=xl.to(dev()), timesteps=ls.to(dev()), x_self_cond=xsc.to(dev()), label=label.to(dev()))
    e = errors(y, ref)
    report(dict(test='cond_lr_step', B=Bl, precision='default', **e))
    assert e['rel_to_max'] < 1e-3, e


def test_full_width_feature_step():
    """configs[4]: obja 3-stage -- the feature net on a shell-8 tree (N8 = 448 232) with the hr net nested as its
    middle (run as_middle, itself without the lr net): octfusion_model_union_3t.py:152-214, graph_unet_hr.py:211-281."""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    doc = shell8_gpu()
    assert doc.total_num == 448232
    net = UNet3DModel(**configs.unet_params('obja_uncond', 'feature'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    x = C.rand_input('fw_feature', doc.total_num, 3)
    log_snr = sampler.beta_linear_log_snr(torch.full((1,), 0.45))
    case = OC.get('feature_step')
    ref, t_or = case['ref'], case['oracle_s']
    for prec, planes, bound in MODES:
        with _Modes(prec, planes):
            y = net(unet_type='feature', x=x.to(dev()), doctree=doc, unet_lr=net.unet_hr,
                    timesteps=log_snr.to(dev()), x_self_cond=None, label=None)
        e = errors(y, ref)
        report(dict(test='feature_step', config='obja_uncond', B=1, N=doc.total_num, precision=prec,
                    planes_kernel=planes, oracle_s=t_or, **e))
        assert e['rel_to_max'] < bound, (prec, planes, e)


# every (depth, Cin, Cout) GraphConv of the three configs (hr / feature nets): K = 7 (Cin + depth - 1)
def layer_shapes():
    from octfusion_amd import configs
    from octfusion_amd.graph_unet_union import UNet3DModel
    from octfusion_amd.modules import GraphConv
    seen = set()
    for cfg, stage in (('snet_uncond', 'hr'), ('snet_cond', 'hr'), ('obja_uncond', 'feature')):
        with torch.device('meta'):
            net = UNet3DModel(**configs.unet_params(cfg, stage))
        for m in net.modules():
            if isinstance(m, GraphConv):
                seen.add((m.n_node_type + 1, m.in_channels, m.out_channels))
    return sorted(seen)


def _sweep_case(d, cin, cout):
    from oracle import modules as OM
    _, o_doc = shell8_oracle()
    N = int(o_doc.graph[d]['keyd'].shape[0])
    w = C.fill_state_dict([('weights', (7 * (cin + d - 1), cout))])['weights']
    x = C.rand_input('sweep_%d_%d_%d' % (d, cin, cout), N, cin)
    return {'ref': OM.graph_conv(x.double(), o_doc, d, w.double(), None, d - 1)}


for _d, _ci, _co in layer_shapes():
    OC.register('sweep_%d_%d_%d' % (_d, _ci, _co), functools.partial(_sweep_case, _d, _ci, _co))


def test_per_layer_sweep():
    """Every distinct (depth, Cin, Cout) GraphConv the three configs contain, at its own graph depth on the
    shell-8 B = 1 tree (depths 4..8), each contraction mode against the oracle in fp64.  modules.py:194-220."""
    from octfusion_amd import modules as M
    doc = shell8_gpu()
    shapes = layer_shapes()
    assert len(shapes) >= 12
    worst = {}
    for d, cin, cout in shapes:
        N = doc.csr(d)[2]
        conv = M.GraphConv(cin, cout, 7, 7, d - 1)
        sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()])
        conv.load_state_dict(sd)
        conv = conv.to(dev())
        x = C.rand_input('sweep_%d_%d_%d' % (d, cin, cout), N, cin)
        ref = OC.get('sweep_%d_%d_%d' % (d, cin, cout))['ref']
        for prec, planes, bound in MODES:
            if prec == 'fp16' and cin % 64:
                continue
            with _Modes(prec, planes):
                from octfusion_amd import ops
                saved = ops.PLANES_MIN_TILES
                ops.PLANES_MIN_TILES = 1               # exercise the planes kernel on every eligible shape
                try:
                    y = conv(x.to(dev()), doc, d, split_input=True)
                finally:
                    ops.PLANES_MIN_TILES = saved
            e = errors(y, ref)
            key = (prec, planes)
            worst[key] = max(worst.get(key, 0.0), e['rel_to_max'])
            report(dict(test='layer', depth=d, N=N, cin=cin, cout=cout, K=7 * (cin + d - 1), precision=prec,
                        planes_kernel=planes, **e))
            assert e['rel_to_max'] < ((2e-5 if prec == 'fp16x3' else 2e-4) if prec != 'fp16' else 5e-3), (d, cin, cout, prec, planes, e)
    report(dict(test='layer_sweep_worst', shapes=len(shapes), **{'%s_%s' % k: v for k, v in worst.items()}))


@pytest.mark.parametrize('persistent', [1, 0])
@pytest.mark.parametrize('prec', ['fp16x3', 'bf16x3', 'fp16'])
def test_planes_kernel_edge_cases(prec, persistent):
    """The LDS-DMA planes GraphConv off the beaten path: ragged batch of 5 with an element that has nothing below the
    full layer (tile / wave boundaries fall inside batch elements: mixed-batch statistics), output widths that are
    not multiples of the 128 / 64-column tiles (200, 72, 64, 132) incl. the clamped-column path, every block geometry
    (128 / 256 rows), bias + time-embedding + residual + fused GroupNorm statistics, planes written by the GroupNorm
    (with the folded aux rows) vs by ofx_planes_split (+ stand-alone pre-pass), M = a handful of rows.  Against the
    oracle in fp64 (modules.py:194-220, 291-314)."""
    from octfusion_amd import _lib, modules as M, ops
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    from oracle import dual_octree as OD, modules as OM, sampler as OS
    mode = {'fp16x3': 3, 'bf16x3': 2, 'fp16': 1}[prec]
    tol = 5e-3 if prec == 'fp16' else 2e-4
    split = C.random_split_small(5, 3, 41, p=0.4)
    split[3] = -1.0
    doc = DualOctree(split2octree_small(split.to(dev()), 5, 3))
    o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 5, 3))
    o_doc.post_processing_for_docnn()
    saved = (ops.get_precision(), ops.PLANES_MIN_TILES)
    ops.set_precision(prec)
    ops.PLANES_MIN_TILES = 1
    _lib.call('ofx_set_gconv_persistent', persistent)       # stream-K blocks / one tile per block (tests/test_gpu_persistent.py)
    try:
        for d, cin, cout, nt, bias in [(5, 64, 200, 4, True), (5, 128, 72, 4, False), (4, 192, 64, 3, True),
                                       (5, 64, 132, 0, False), (3, 64, 128, 2, True)]:
            conv = M.GraphConv(cin, cout, 7, 7, nt, use_bias=bias)
            gn = M.DualOctreeGroupNorm(cin)
            sd = C.fill_state_dict([('c.' + k, tuple(v.shape)) for k, v in conv.state_dict().items()] +
                                   [('g.' + k, tuple(v.shape)) for k, v in gn.state_dict().items()])
            conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('c.')})
            gn.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith('g.')})
            conv, gn = conv.to(dev()), gn.to(dev())
            N = doc.csr(d)[2]
            x = C.rand_input('edge_x_%d_%d' % (d, cin), N, cin)
            emb = C.rand_input('edge_e_%d' % cout, 5, cout)
            res = C.rand_input('edge_r_%d_%d' % (d, cout), N, cout)
            h_ref = OM.silu(OM.dual_octree_group_norm(x.double(), o_doc, d, sd['g.weights'].double(), sd['g.bias'].double()))
            ref = OM.graph_conv(h_ref, o_doc, d, sd['c.weights'].double(),
                                sd['c.bias'].double() if bias else None, nt) + emb.double()[o_doc.batch_id(d)] + res.double()
            outs = []
            for tile in (2, 4):
                _lib.call('ofx_set_gconv2_tile', tile)
                # (a) planes + aux rows written by the GroupNorm launch
                hp = gn(x.to(dev()), doc, d, act='silu', planes=mode)
                assert ops.planes_of(hp) == mode and getattr(hp, ops.AUX_ATTR, None) is not None
                with ops.stats_scope(dev()):
                    conv.emit_stats = True
                    stats = ops.stats_zeros(5 * cout * 2, dev()) if cout % 4 == 0 else None
                    pw2 = conv._pw2.get(conv.weights, cin, nt if nt > 1 else 0, mode)
                    seg_ptr, col, _, _ = doc.csr(d)
                    y = ops.graphconv_planes(hp, mode, seg_ptr, col, doc.ext(d), pw2, cin, nt if nt > 1 else 0,
                                             doc.type_frac_planes(d, nt, mode) if nt > 1 else None,
                                             conv.bias if bias else None, emb.to(dev()), doc.batch_id32(d), res.to(dev()),
                                             None, stats=stats)
                    e = errors(y, ref)
                    assert e['rel_to_max'] < tol, (d, cin, cout, tile, e)
                    if stats is not None:
                        bid = doc.batch_id32(d).long()
                        want = torch.zeros(5, cout, 2, dtype=torch.float64, device=dev())
                        want[:, :, 0].index_add_(0, bid, y.double())
                        want[:, :, 1].index_add_(0, bid, y.double() ** 2)
                        assert float((stats.view(5, cout, 2) - want).abs().max()) <= 1e-5 * float(want.abs().max())
                # (b) the same activation converted by ofx_planes_split: stand-alone pre-pass instead of folded aux rows
                hf = gn(x.to(dev()), doc, d, act='silu')
                y2 = conv(ops.planes_split(hf, mode), doc, d, emb=emb.to(dev()), res=res.to(dev()))
                assert errors(y2, ref)['rel_to_max'] < tol
                outs.append(y)
            if persistent:      # tiles cut by a share boundary are summed in two groups: same arithmetic, other rounding
                assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(outs[0].abs().max())
                assert not ops.sync_error(dev())
            else:
                assert torch.equal(outs[0], outs[1]), 'geometries disagree bit-wise'
            report(dict(test='planes_edge', precision=prec, depth=d, N=N, cin=cin, cout=cout, nt=nt, **errors(outs[0], ref)))
        # a graph level with a handful of rows (depth-3 full layer of ONE tiny tree: 512 rows < one 256-row tile x 2)
        tiny = C.random_split_small(1, 2, 7, p=0.5)
        doc1 = DualOctree(split2octree_small(tiny.to(dev()), 4, 2))
        o1 = OD.OracleDualOctree(OS.split2octree_small(tiny, 4, 2))
        o1.post_processing_for_docnn()
        conv = M.GraphConv(64, 64, 7, 7, 2).to(dev())
        x = C.rand_input('edge_tiny', doc1.csr(2)[2], 64)
        ref = OM.graph_conv(x.double(), o1, 2, conv.weights.detach().cpu().double(), None, 2)
        y = conv(x.to(dev()), doc1, 2, split_input=True)
        assert errors(y, ref)['rel_to_max'] < tol
    finally:
        _lib.call('ofx_set_gconv2_tile', 0)
        _lib.call('ofx_set_gconv_persistent', 1)
        ops.set_precision(saved[0])
        ops.PLANES_MIN_TILES = saved[1]
