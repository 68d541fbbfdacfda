"""How much rounding noise does the REFERENCE's own fp32 arithmetic carry on a whole denoising step?
The oracle (the reference's op sequence, oracle/) is run twice on the same weights and inputs at the real network
widths: in float32 (what the reference computes) and in float64 (the same ops, exact for this purpose).  The
difference is the floor any fp32-class implementation is measured against; an element-wise tolerance below it
cannot be met by the reference itself.  CPU only.

    python tools/oracle_noise.py [--out profiles/r03/oracle_fp32_noise.json]
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
from octfusion_amd import configs, synthetic
from octfusion_amd.graph_unet_union import UNet3DModel
from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU

torch.set_grad_enabled(False)


def figures(a, b):
    d = (a.double() - b.double()).abs()
    scale = float(b.abs().max())
    e = d / b.double().abs().clamp(min=1e-2 * scale)
    return dict(rel_to_max=float(d.max()) / scale, elementwise_p999=float(torch.quantile(e.flatten()[:4_000_000], 0.999)),
                elementwise_max=float(e.max()))


def dbl(parts):
    return {k: {kk: (vv.double() if vv.is_floating_point() else vv) for kk, vv in v.items()} for k, v in parts.items()}


ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
ap.add_argument('--batch', type=int, default=2)
args = ap.parse_args()
rows = []
B = args.batch
split = synthetic.shell6_split(B, jitter=True)
o_doc = OD.OracleDualOctree(OS.split2octree_small(split, 6, 4))
o_doc.post_processing_for_docnn()
for cfgname in ('snet_uncond', 'snet_cond'):
    net = UNet3DModel(**configs.unet_params(cfgname, 'hr'))
    sd = synthetic.random_state_dict(net)
    st = configs.stage_cfgs(cfgname)
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr')}
    p64 = dbl(parts)
    x = C.rand_input('fw_' + cfgname, o_doc.total_num, 3)
    log_snr = OS.beta_linear_log_snr(torch.full((B,), 0.6))
    label = (torch.arange(B) % 5) if st['hr'].get('num_classes') else None
    t0 = time.time()
    r32 = OU.hr_forward(parts['unet_hr'], st['hr'], x, o_doc, log_snr, label, parts['unet_lr'], st['lr'])
    t32 = time.time() - t0
    with OM.working_float(torch.float64):
        r64 = OU.hr_forward(p64['unet_hr'], st['hr'], x.double(), o_doc, log_snr.double(), label, p64['unet_lr'], st['lr'])
    rows.append(dict(step='hr', config=cfgname, B=B, N=o_doc.total_num, fp32_s=t32, **figures(r32, r64)))
    print(json.dumps(rows[-1]))
    if cfgname == 'snet_uncond':
        Bl = 4
        xl = C.rand_input('fw_lr', Bl, 8, 16, 16, 16)
        xsc = C.rand_input('fw_lr_sc', Bl, 8, 16, 16, 16)
        ls = OS.beta_linear_log_snr(torch.full((Bl,), 0.3))
        r32 = OU.lr_forward(parts['unet_lr'], st['lr'], xl, ls, xsc, None)
        with OM.working_float(torch.float64):
            r64 = OU.lr_forward(p64['unet_lr'], st['lr'], xl.double(), ls.double(), xsc.double(), None)
        rows.append(dict(step='lr', config=cfgname, B=Bl, **figures(r32, r64)))
        print(json.dumps(rows[-1]))
if args.out:
    json.dump(dict(what='fp32 oracle vs the same op sequence in fp64 (tools/oracle_noise.py): the reference\'s own rounding '
                        'noise on one whole denoising step at the real widths; element-wise figure = |a-b| / max(|b|, 1% of '
                        'max|b|)', rows=rows), open(args.out, 'w'), indent=1)
