"""Generate-style driver: independent shapes sharded over the ranks of one node, end to end.

Mirror of the reference's generation loop (train.py:166-185 -> OctFusionModel.sample,
octfusion_model_union.py:354-401): lr DDIM loop -> octree -> hr DDIM loop (-> feature loop for the 3-stage model) ->
GraphVAE.decode_code -> NeuralMPU SDF on the resolution^3 lattice (get_sdfs, :425-433).  Marching cubes / mesh export
(:435-468, skimage + trimesh on the host) is outside the device path.

* The sampling nets hold the EMA weights, as the reference's generate does (train.py:181 ``ema=True``;
  ``self.ema_df`` and ``unet_lr=self.ema_df.unet_lr``, octfusion_model_union.py:319,391).
* Rank r of W produces the shapes with ``result_index = i * W + r`` (dist.shard_indices), ``--batch`` of them per
  call: shapes are independent, so they are batched -- and every shape of a batch still gets exactly the noise the
  reference's one-shape-per-call loop would draw for its result index (CascadeSampler.sample(shape_indices=)).
* Rank 0 owns the weights (checkpoints, or seeded random weights when none are given) and broadcasts U-Net AND VAE
  once over RCCL (dist.broadcast_module_); nothing is communicated per step.

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


def build_models(config, ckpt=None, vae_ckpt=None, with_vae=True, rank=0, allow_pickle=False):
    """(net with EMA weights, vae or None) on the CPU; only rank 0 reads files / draws the synthetic weights."""
    cfg = configs.CONFIGS[config]
    stage = cfg['unet_type'][-1]
    net = UNet3DModel(**configs.unet_params(config, stage))
    vae = None
    if with_vae:
        from .graph_vae import GraphVAE
        vae = GraphVAE(**configs.vae_params(config))
    if rank == 0:
        if ckpt:
            # `net` plays both roles of load_ckpt: the df_* weights are loaded first and then overwritten by ema_df_*,
            # so a file whose two sets differ leaves the EMA set in the sampling net
            checkpoint.load_ckpt(ckpt, net, ema_df=net, allow_pickle=allow_pickle)
        else:
            net.load_state_dict(synthetic.random_state_dict(net))
        if vae is not None:
            if vae_ckpt:
                checkpoint.load_vae(vae_ckpt, vae, allow_pickle=allow_pickle)
            else:
                vae.load_state_dict(synthetic.random_state_dict(vae))
    return net, vae


def plan(n_shapes, rank, world, shapes_per_call):
    """The groups of result indices rank `rank` generates, in order: its share {i : i mod world == rank}
    (train.py:168) cut into calls of `shapes_per_call`."""
    mine = dist.shard_indices(n_shapes, rank, world)
    return [mine[g0:g0 + shapes_per_call] for g0 in range(0, len(mine), shapes_per_call)]


def prepare(config, rank, device, ckpt=None, vae_ckpt=None, with_vae=True, allow_pickle=False):
    """(net, vae, bytes broadcast): models on `device` with rank 0's weights on every rank -- ONE flat broadcast per
    model, U-Net and VAE."""
    net, vae = build_models(config, ckpt, vae_ckpt, with_vae=with_vae, rank=rank, allow_pickle=allow_pickle)
    net = net.to(device).eval()
    nbytes = dist.broadcast_module_(net, src=0)
    if vae is not None:
        vae = vae.to(device).eval()
        nbytes += dist.broadcast_module_(vae, src=0)
    return net, vae, nbytes


def generate(net, cfg, n_shapes, rank, world, seed=0, ddim_steps=200, label=None, vae=None, out_dir=None,
             shapes_per_call=1, use_graph=None, sdf_resolution=None, timings=None):
    """Yields (result indices, output dict, seconds) for every group of shapes this rank owns."""
    cs = CascadeSampler(net, cfg, vae)
    dev = cs.device
    for idxs in plan(n_shapes, rank, world, shapes_per_call):
        lab = None
        if label is not None:
            lab = torch.full((len(idxs),), int(label), dtype=torch.long, device=dev)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = cs.sample(len(idxs), ddim_steps=ddim_steps, label=lab, seed=seed, shape_indices=idxs,
                        use_graph=use_graph, sdf_resolution=sdf_resolution if vae is not None else None,
                        timings=timings)
        if dev.type == 'cuda':
            torch.cuda.synchronize()
            from . import ops
            ops.raise_on_sync_error(dev)         # (the DDIM loops check per stage; this covers the VAE decode)
        dt = time.perf_counter() - t0
        if out_dir is not None:
            write_outputs(out_dir, idxs, out, cfg)
        yield idxs, out, dt


def write_outputs(out_dir, idxs, out, cfg):
    """Per shape: <index>/split_small.pth (+ split_large.pth) in the reference's sample-file format
    (tools/gen_split.py:50-54), and <index>/sdf.pt when the SDF lattice was computed."""
    from .octree import octree2split_large, octree2split_small
    small = octree2split_small(out['octree_small'], cfg['full_depth'])
    large = bid = None
    if 'octree_large' in out:
        # depth-8 cascade (3-stage model): the stage-2 sample format, one [nnum6 of the shape, 8] tensor per shape
        # (gen_split.py:50-54 writes it from a one-shape octree; here the batch is sliced by the depth-6 batch ids)
        sd = cfg['input_depth'][1]
        large = octree2split_large(out['octree_large'], sd)
        bid = out['octree_large'].batch_id(sd)
    for b, i in enumerate(idxs):
        d = os.path.join(out_dir, str(i))
        os.makedirs(d, exist_ok=True)
        torch.save(small[b].cpu(), os.path.join(d, 'split_small.pth'))
        if large is not None:
            torch.save(large[bid == b].cpu(), os.path.join(d, 'split_large.pth'))
        if 'sdfs' in out:
            torch.save(out['sdfs'][b].cpu(), os.path.join(d, 'sdf.pt'))


def run(args, rank, local_rank, world, device):
    cfg = configs.CONFIGS[args.config]
    net, vae, nbytes = prepare(args.config, rank, device, args.ckpt, args.vae, with_vae=not args.no_vae,
                               allow_pickle=getattr(args, 'allow_pickle', False))
    label = args.category if cfg.get('num_classes') else None
    if cfg.get('num_classes') and label is None:
        label = 0
    per_rank = len(dist.shard_indices(args.shapes, rank, world))
    batch = args.batch or max(1, min(8, per_rank))
    timings = {}
    done = []
    for idxs, out, dt in generate(net, cfg, args.shapes, rank, world, args.seed, args.steps, label, vae, args.out, batch,
                                  sdf_resolution=args.sdf_resolution, timings=timings):
        done.append((idxs, dt))
    total = sum(dt for _, dt in done)
    tmax = dist.max_over_ranks(total, device)
    res = {'config': args.config, 'shapes': args.shapes, 'world': world, 'steps_per_stage': args.steps,
           'shapes_per_call': batch, 'weight_broadcast_bytes': nbytes, 'seconds_max_over_ranks': tmax,
           'seconds_per_shape': tmax * world / args.shapes if args.shapes else None,
           'shapes_per_s': args.shapes / tmax if tmax > 0 else None, 'rank0_indices': [i for g, _ in done for i in g],
           'rank0_phase_seconds': timings, 'sdf_resolution': args.sdf_resolution if vae is not None else None}
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='snet_uncond', choices=sorted(configs.CONFIGS))
    ap.add_argument('--shapes', type=int, default=8)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--batch', type=int, default=None, help='shapes per sample() call (default: min(8, shapes per rank))')
    ap.add_argument('--category', type=int, default=None, help='class label for the conditional config')
    ap.add_argument('--ckpt', default=None, help='df_*.pth (the EMA weights are used, as the reference does)')
    ap.add_argument('--vae', default=None, help='GraphVAE checkpoint (seeded random weights when absent)')
    ap.add_argument('--no-vae', action='store_true', help='stop after the last DDIM stage (no decode, no SDF)')
    ap.add_argument('--allow-pickle', action='store_true',
                    help='read the checkpoint files with the full unpickler (only for files you trust)')
    ap.add_argument('--sdf-resolution', type=int, default=256)
    ap.add_argument('--out', default=None)
    args = ap.parse_args(argv)
    rank, local_rank, world = dist.init()
    from . import _lib
    _lib.require_device()
    torch.cuda.set_device(local_rank)
    res = run(args, rank, local_rank, world, torch.device('cuda', local_rank))
    if rank == 0:
        print(json.dumps(res))
    dist.barrier()
    return res


if __name__ == '__main__':
    main()
