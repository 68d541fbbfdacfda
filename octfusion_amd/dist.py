"""One process per GPU: batch-shard of independent shapes + ONE weight broadcast.

The reference samples independent shapes per rank and has every rank read the
checkpoint from disk (train.py:166-185, octfusion_model_union.py:525-545).  Here
rank 0 owns the weights and broadcasts them once, flattened into a single
fp32 buffer, over RCCL/xGMI (132-335 MB: one large collective instead of one
per tensor); there is no communication inside a denoising step.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'      # "nccl" is RCCL on ROCm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_indices(n_items, rank, world):
    """Rank r takes items {i : i mod world == r} -- the reference's rule (train.py:168)."""
    return list(range(rank, n_items, world))


@torch.no_grad()
def broadcast_module_(module, src=0):
    """Broadcast every parameter and buffer of `module` from `src` as ONE flat fp32 buffer."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    # the parameters themselves (not .data, which has its own version counter): copy_ below bumps _version, which is
    # what the packed-weight caches key on -- a forward that ran before the broadcast cannot leave stale packs
    tensors = list(module.parameters()) + list(module.buffers())
    tensors = [t for t in tensors if t.is_floating_point()]
    if not tensors:
        return 0
    dev = tensors[0].device
    total = sum(t.numel() for t in tensors)
    flat = torch.empty(total, dtype=torch.float32, device=dev)
    off = 0
    if dist.get_rank() == src:
        for t in tensors:
            flat[off:off + t.numel()].copy_(t.reshape(-1))
            off += t.numel()
    dist.broadcast(flat, src=src)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    return total * 4


@torch.no_grad()
def all_reduce_mean_(grads, bucket_bytes=256 << 20):
    """Data-parallel training (the reference wraps its nets in DistributedDataParallel: octfusion_model_union.py:185-196,
    octfusion_model_vae.py:121-130): average the gradient dict over the ranks in place.  Gradients are packed, in
    sorted key order, into flat fp32 buckets of up to `bucket_bytes` -- a few large ring all-reduces (per-link bound on
    xGMI, so size matters more than count) instead of one per tensor.  Every rank must hold the same keys; a key that
    is missing on some rank (a parameter without gradient there) is an error the caller has to resolve.
    Returns the number of bytes reduced."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    keys = sorted(grads)
    # ranks with different key sets / sizes would pack different flat buckets: an RCCL hang or silent corruption.
    # One tiny all-reduce of (key count, total elements, a key-name checksum) turns that into an error.
    import zlib
    dev = grads[keys[0]].device if keys else torch.device('cpu')
    sig = torch.tensor([len(keys), sum(grads[k].numel() for k in keys),
                        sum(zlib.crc32(k.encode()) for k in keys) % (1 << 40)], dtype=torch.int64, device=dev)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi):
        raise RuntimeError('all_reduce_mean_: the ranks hold different gradient sets (keys / sizes: min %s, max %s) -- a '
                           'parameter without gradient on some rank must be zero-filled by the caller'
                           % (lo.tolist(), hi.tolist()))
    total = 0
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size, total
        if not bucket:
            return
        flat = torch.cat([grads[k].reshape(-1).float() for k in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world
        off = 0
        for k in bucket:
            n = grads[k].numel()
            grads[k].copy_(flat[off:off + n].view_as(grads[k]))
            off += n
        total += flat.numel() * 4
        bucket, size = [], 0
    for k in keys:
        nb = grads[k].numel() * 4
        if bucket and size + nb > bucket_bytes:
            flush()
        bucket.append(k)
        size += nb
    flush()
    return total


def max_over_ranks(value, device):
    """MAX all-reduce of a python float (timing)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device):
    """[value of rank 0, ..., value of rank W-1] on every rank (per-rank step times)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
