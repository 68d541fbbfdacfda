"""Generate-style driver: independent shapes sharded over the ranks of one node.

Mirror of the reference's generation loop (train.py:166-185): rank r of W produces the shapes with
``result_index = iter_i * W + r`` (octfusion_amd.dist.shard_indices), one shape per call of ``sample`` with
``batch_size`` 1 by default, seeded per shape the way the reference seeds them
(octfusion_model_union.py:372,390 via CascadeSampler.sample(seed=, save_index=)).  Rank 0 owns the weights (a
checkpoint, or seeded random weights when none is given) and broadcasts them once over RCCL
(dist.broadcast_module_); nothing is communicated per step.

    python -m octfusion_amd.generate --config snet_uncond --shapes 8 --steps 200 [--ckpt df.pth --vae vae.pth]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        -m octfusion_amd.generate --config snet_cond --shapes 32 --category 2
"""
import argparse
import json
import os
import time

import torch

from . import checkpoint, configs, dist, synthetic
from .graph_unet_union import UNet3DModel
from .pipeline import CascadeSampler


def generate(net, cfg, n_shapes, rank, world, seed=0, ddim_steps=200, label=None, vae=None, out_dir=None,
             shapes_per_call=1, use_graph=None, sdf_resolution=None):
    """Yields (result_index, output dict, seconds) for every shape group this rank owns."""
    cs = CascadeSampler(net, cfg, vae)
    dev = cs.device
    for result_index in dist.shard_indices(n_shapes, rank, world):
        lab = None
        if label is not None:
            lab = torch.full((shapes_per_call,), int(label), dtype=torch.long, device=dev)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cs.sample(shapes_per_call, ddim_steps=ddim_steps, label=lab, seed=seed, save_index=result_index,
                        use_graph=use_graph, sdf_resolution=sdf_resolution)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if out_dir is not None:
            tree = out.get('octree_large', out['octree_small'])
            checkpoint.write_splits(os.path.join(out_dir, str(result_index)), tree, cfg['full_depth'],
                                    cfg['input_depth'][1])
        yield result_index, out, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='snet_uncond', choices=sorted(configs.CONFIGS))
    ap.add_argument('--shapes', type=int, default=8)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--category', type=int, default=None, help='class label for the conditional config')
    ap.add_argument('--ckpt', default=None)
    ap.add_argument('--vae', default=None)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    rank, local_rank, world = dist.init()
    from . import _lib
    _lib.require_device()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    cfg = configs.CONFIGS[args.config]
    stage = cfg['unet_type'][-1]
    net = UNet3DModel(**configs.unet_params(args.config, stage))
    if rank == 0:
        if args.ckpt:
            checkpoint.load_ckpt(args.ckpt, net, None)
        else:
            net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev).eval()
    nbytes = dist.broadcast_module_(net, src=0)
    label = args.category if cfg.get('num_classes') else None
    if cfg.get('num_classes') and label is None:
        label = 0
    done = []
    for idx, out, dt in generate(net, cfg, args.shapes, rank, world, args.seed, args.steps, label, None, args.out):
        done.append((idx, dt))
    tmax = dist.max_over_ranks(sum(dt for _, dt in done), dev)
    if rank == 0:
        print(json.dumps({'shapes': args.shapes, 'world': world, 'steps_per_stage': args.steps,
                          'weight_broadcast_bytes': nbytes, 'seconds_max_over_ranks': tmax,
                          'shapes_per_s': args.shapes / tmax if tmax > 0 else None,
                          'rank0_shapes': [i for i, _ in done]}))
    dist.barrier()


if __name__ == '__main__':
    main()
