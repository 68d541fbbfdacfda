"""Where does a whole denoising step lose accuracy, and what does each remedy cost?  (GPU box.)

One step of the hr net (+ nested lr) and of the lr net at the real widths, under four settings of the contraction
precision, against the CPU oracle run in float32 (the reference's arithmetic) AND in float64 (the same op sequence,
exact for this purpose):
    fp16x3 (default): fp16 hi + lo operand pairs, three fp16 MFMAs per product
    pure_bf16x3     : bf16 pairs everywhere (round 2's default)
    + dense net fp32: bf16 pairs, the dense lr net in exact fp32          (ops.POLICY['dense_net'])
    + small layers  : + GEMMs / GraphConvs with <= 64 channels exact fp32  (ops.POLICY['small_gemm'])
    fp32            : everything exact fp32 (ofx_set_precision(1))
Figures: rel-to-max, element-wise p99.9 / max with the 1 % floor (tests/test_gpu_fullwidth.errors), eager ms per step.

    python tools/precision_attribution.py --out gpurun_out/precision_attribution.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import torch

import common as C
from octfusion_amd import configs, ops, synthetic
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.graph_unet_union import UNet3DModel
from octfusion_amd.octree import split2octree_small
from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
SETTINGS = [('fp16x3 (default)', 'fp16x3', dict(dense_net=None, small_gemm=None)),
            ('pure_bf16x3', 'bf16x3', dict(dense_net=None, small_gemm=None)),
            ('bf16x3 + dense net fp32', 'bf16x3', dict(dense_net='fp32', small_gemm=None)),
            ('bf16x3 + dense net and small layers fp32', 'bf16x3', dict(dense_net='fp32', small_gemm='fp32')),
            ('fp32', 'fp32', dict(dense_net=None, small_gemm=None))]


def figures(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max())
    d = (a - b).abs()
    e = d / b.abs().clamp(min=1e-2 * scale)
    return dict(rel_to_max=float(d.max()) / scale, elementwise_p999=float(torch.quantile(e.flatten()[:4_000_000], 0.999)),
                elementwise_max=float(e.max()))


def dbl(parts):
    return {k: {kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in v.items()} for k, v in parts.items()}


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
args = ap.parse_args()
rows = []
saved_policy = dict(ops.POLICY)
B = 2
split = synthetic.shell6_split(B, jitter=True)
doc = DualOctree(split2octree_small(split.to(dev), 6, 4))
o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 6, 4))
o_doc.post_processing_for_docnn()
for cfgname in ('snet_uncond', 'snet_cond'):
    net = UNet3DModel(**configs.unet_params(cfgname, 'hr'))
    sd = synthetic.random_state_dict(net)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    st = configs.stage_cfgs(cfgname)
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr')}
    p64 = dbl(parts)
    x = C.rand_input('fw_' + cfgname, doc.total_num, 3)
    log_snr = OS.beta_linear_log_snr(torch.full((B,), 0.6))
    label = (torch.arange(B) % 5) if st['hr'].get('num_classes') else None
    r32 = OU.hr_forward(parts['unet_hr'], st['hr'], x, o_doc, log_snr, label, parts['unet_lr'], st['lr'])
    with OM.working_float(torch.float64):
        r64 = OU.hr_forward(p64['unet_hr'], st['hr'], x.double(), o_doc, log_snr.double(), label, p64['unet_lr'], st['lr'])
    rows.append(dict(step='hr', config=cfgname, setting='oracle_fp32_vs_fp64', **figures(r32, r64)))
    print(json.dumps(rows[-1]))
    cases = [('hr', lambda: net(unet_type='hr', x=x.to(dev), doctree=doc, unet_lr=net.unet_lr, timesteps=log_snr.to(dev),
                                x_self_cond=None, label=label.to(dev) if label is not None else None), r32, r64)]
    if cfgname == 'snet_uncond':
        Bl = 4
        xl = C.rand_input('fw_lr', Bl, 8, 16, 16, 16)
        xsc = C.rand_input('fw_lr_sc', Bl, 8, 16, 16, 16)
        ls = OS.beta_linear_log_snr(torch.full((Bl,), 0.3))
        l32 = OU.lr_forward(parts['unet_lr'], st['lr'], xl, ls, xsc, None)
        with OM.working_float(torch.float64):
            l64 = OU.lr_forward(p64['unet_lr'], st['lr'], xl.double(), ls.double(), xsc.double(), None)
        rows.append(dict(step='lr', config=cfgname, setting='oracle_fp32_vs_fp64', **figures(l32, l64)))
        print(json.dumps(rows[-1]))
        cases.append(('lr', lambda: net(unet_type='lr', x=xl.to(dev), timesteps=ls.to(dev), x_self_cond=xsc.to(dev)), l32, l64))
    for step, run, ref32, ref64 in cases:
        for name, prec, pol in SETTINGS:
            ops.POLICY.update(pol)
            ops.set_precision(prec)
            try:
                y = run()
                ms = timeit(run)
            finally:
                ops.set_precision(ops.DEFAULT_PRECISION)
                ops.POLICY.update(saved_policy)
            rows.append(dict(step=step, config=cfgname, setting=name, eager_ms=ms,
                             vs_fp32_oracle=figures(y, ref32), vs_fp64_oracle=figures(y, ref64)))
            print(json.dumps(rows[-1]))
if args.out:
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(dict(what=__doc__, rows=rows), open(args.out, 'w'), indent=1)
