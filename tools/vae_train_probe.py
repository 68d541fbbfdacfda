"""Timing of the training-side paths added in round 2 (run on the GPU box):
point cloud -> octree build (points/s), dual-graph build (fresh vs incremental), VAE training step (ms/step) at the
reference's VAE configuration (configs/vae_snet_train.yaml: depth 8, full_depth 4, depth_stop 6, resblk_num 2).

    python tools/vae_train_probe.py [--batch 4] [--points 200000] [--pos 20000] [--json out.json]
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import synthetic, training as T, vae_training as VT
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.graph_vae import GraphVAE
from octfusion_amd.octree import Points, build_octree_batch

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--points', type=int, default=200000)
ap.add_argument('--pos', type=int, default=20000)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--json', default=None)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)


def sphere(n, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(n, 3, generator=g)
    nrm = v / v.norm(dim=1, keepdim=True)
    return (nrm * (0.5 + 0.03 * seed) + 0.02 * seed).clamp(-1, 1).contiguous(), nrm.contiguous()


def timed(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return out, 1e3 * (time.perf_counter() - t0) / reps


clouds = [Points(*(t.to(dev) for t in sphere(args.points, i))) for i in range(args.batch)]
res = {'batch': args.batch, 'points_per_shape': args.points, 'pos_per_shape': args.pos}
oc, res['octree_build_first_ms'] = timed(lambda: build_octree_batch(clouds, 8, 4))
oc, res['octree_build_ms'] = timed(lambda: build_octree_batch(clouds, 8, 4), 5)
res['points_per_s'] = args.batch * args.points / (res['octree_build_ms'] * 1e-3)
res['nnum'] = [int(v) for v in oc.nnum]
doc, res['dual_graph_first_ms'] = timed(lambda: DualOctree(oc))
doc, res['dual_graph_ms'] = timed(lambda: DualOctree(oc), 5)
res['graph_nodes_depth8'] = doc.total_num

vae = GraphVAE(depth=8, channel_in=4, nout=4, full_depth=4, depth_stop=6, depth_out=8, resblk_type='basic',
               resblk_num=2, embed_dim=3)
vae.load_state_dict(synthetic.random_state_dict(vae))
vae = vae.to(dev)
data = doc.get_input_feature()
g = torch.Generator().manual_seed(3)
n_pos = args.batch * args.pos
pos = torch.cat([torch.rand(n_pos, 3, generator=g) * 1.6 - 0.8,
                 torch.arange(args.batch).repeat_interleave(args.pos).float().unsqueeze(1)], 1).to(dev)
sdf_gt = (torch.randn(n_pos, generator=g) * 0.05).to(dev)
grad_gt = torch.nn.functional.normalize(torch.randn(n_pos, 3, generator=g), dim=1).to(dev)
opt = T.AdamW(vae.named_parameters(), lr=1e-5)
noise = torch.randn(doc.csr(6)[2], 3, generator=g).to(dev)
step = lambda: VT.vae_stage_step(vae, opt, data, doc, doc, pos, sdf_gt, grad_gt, noise, 0.1)     # noqa: E731
losses, res['vae_step_first_ms'] = timed(step)
losses, res['vae_step_ms'] = timed(step, args.steps)
res['loss'] = float(losses['loss'])
assert math.isfinite(res['loss'])
# growth path of the decoder: fresh rebuild per depth vs incremental adoption
code = vae.encode(data, doc, sample=False)[0]
_, res['vae_decode_grow_ms'] = timed(lambda: vae.decode_code(code, doc, update_octree=True), 3)
print(json.dumps(res))
if args.json:
    json.dump(res, open(args.json, 'w'), indent=1)
