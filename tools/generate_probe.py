"""seconds per generated shape, end to end, with the phase split (GPU box): the generate driver at the real widths of
snet_uncond with synthetic weights, B = 1 and B = 8 shapes per call, 200 DDIM steps per stage, SDF at 256^3.

    python tools/generate_probe.py --out gpurun_out/generate_probe.json [--steps 200]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from octfusion_amd import configs, generate as G, synthetic
from octfusion_amd.pipeline import CascadeSampler

ap = argparse.ArgumentParser()
ap.add_argument('--out', default=None)
ap.add_argument('--steps', type=int, default=200)
ap.add_argument('--config', default='snet_uncond')
ap.add_argument('--batches', default='1,8')
args = ap.parse_args()
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
cfg = configs.CONFIGS[args.config]
net, vae, _ = G.prepare(args.config, 0, dev)
cs = CascadeSampler(net, cfg, vae)
cs_lr = CascadeSampler(net, dict(cfg, unet_type=cfg['unet_type'][:1], df_type=cfg['df_type'][:1]), None)    # the lr loop alone
rows = []
# synthetic weights make the lr stage emit noise-like split codes; the bench shapes (shell-6) are fed instead so that
# the hr stage, the decoder and the SDF sweep see ShapeNet-sized octrees -- the lr stage is still run and timed
for B in [int(b) for b in args.batches.split(',')]:
    split = synthetic.shell6_split(B, jitter=True).to(dev)
    for rep in range(2):                                   # first pass: weight packing, lazy code loading, graph capture
        t_lr = {}
        cs_lr.sample(B, ddim_steps=args.steps, seed=0, shape_indices=list(range(B)), timings=t_lr)
        tim = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cs.sample(B, ddim_steps=args.steps, seed=0, shape_indices=list(range(B)), split_small=split,
                        sdf_resolution=256, timings=tim)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tim['lr_steps'] = t_lr['lr_steps']
    total = dt + t_lr['lr_steps']
    rows.append(dict(config=args.config, shapes_per_call=B, steps_per_stage=args.steps, nodes_depth6=int(out['doctree'].total_num),
                     seconds_per_call=total, seconds_per_shape=total / B, phase_seconds=tim,
                     ms_per_step={'lr': 1e3 * tim['lr_steps'] / args.steps, 'hr': 1e3 * tim['hr_steps'] / args.steps},
                     note='second pass (weights packed, code loaded); shell-6 split codes fed to the hr stage (synthetic '
                          'weights make the lr stage emit noise-like codes), the lr loop timed on its own'))
    print(json.dumps(rows[-1]))
if args.out:
    os.makedirs(os.path.dirname(args.out) or '.', exist_ok=True)
    json.dump(dict(what=__doc__, rows=rows), open(args.out, 'w'), indent=1)
