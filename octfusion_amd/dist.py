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


def _active(force=False):
    """Is there a process group whose collectives should run?  World size 1 short-circuits every helper below unless
    `force` (or a group that was created with init(force=True)): then the one-rank collective really goes through the
    backend -- RCCL on a GPU box -- which is how the device path of this module is exercised on a single MI355X
    (tests/test_gpu_dist.py, `bench.py --gpus 1`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or force or _FORCED


_FORCED = False


def init(backend=None, force=False):
    """Initialise torch.distributed from the torchrun environment (no-op for world size 1 unless `force`: then a
    one-rank group is created on 127.0.0.1 and every helper of this module runs its collective through it)."""
    global _FORCED
    rank, local_rank, world = env_world()
    if force and world == 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        import socket
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group(backend=backend, init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1,
                                **_device_kw(backend, local_rank))
        _FORCED = True
        return rank, local_rank, world
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'      # "nccl" is RCCL on ROCm
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **_device_kw(backend, local_rank))
    return rank, local_rank, world


def _device_kw(backend, local_rank):
    """RCCL: bind the group to this rank's GPU at creation (`device_id`) -- the communicator is then created eagerly, on
    the right device, instead of lazily by the first collective on "the device under the current context" (torch's
    warning in the round-4 driver log; with 8 ranks a rank whose current device is still 0 at that moment would put two
    ranks on one GPU)."""
    if backend != 'nccl':
        return {}
    torch.cuda.set_device(local_rank)
    return {'device_id': torch.device('cuda', local_rank)}


def shard_indices(n_items, rank, world):
    """Rank r takes items {i : i mod world == r} -- the reference's rule (train.py:168)."""
    return list(range(rank, n_items, world))


@torch.no_grad()
def broadcast_module_(module, src=0, force=False):
    """Broadcast every parameter and buffer of `module` from `src` as ONE flat fp32 buffer."""
    if not _active(force):
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


def _pg_device():
    """Device a collective's tensors must live on for the default process group (RCCL: this rank's GPU)."""
    if dist.get_backend() == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return torch.device('cpu')


_SIGNATURES = {}


def _check_gradient_set(grads, keys):
    """Ranks with different key sets / sizes would pack different flat buckets: an RCCL hang or silent corruption.
    ONE tiny all-reduce -- MAX over [sig, -sig], i.e. max and min at once -- of (key count, total elements, a checksum
    over every `name:numel` pair) turns that into an error.  EVERY rank issues it on EVERY call (a skip
    decided from rank-local state is itself a collective mismatch the moment the sets diverge -- the rank whose set
    changed would all-reduce six int64 words against its peers' first fp32 bucket); only the signature of a set seen
    before is cached.  Cost per training step: one 48-byte collective + one host read, against a 40-300 ms step."""
    import zlib
    ident = tuple((k, grads[k].numel()) for k in keys)
    sig = _SIGNATURES.get(ident)
    if sig is None:
        sig = [len(keys), sum(n for _, n in ident), sum(zlib.crc32(('%s:%d' % kn).encode()) for kn in ident) % (1 << 40)]
        if len(_SIGNATURES) < 64:
            _SIGNATURES[ident] = sig
    t = torch.tensor(sig + [-v for v in sig], dtype=torch.int64, device=_pg_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    hi, lo = t[:3].tolist(), [-v for v in t[3:].tolist()]
    if hi != lo:
        raise RuntimeError('all_reduce_mean_: the ranks hold different gradient sets (keys / sizes: min %s, max %s) -- a '
                           'parameter without gradient on some rank must be zero-filled by the caller' % (lo, hi))


@torch.no_grad()
def all_reduce_mean_(grads, bucket_bytes=256 << 20, force=False):
    """Data-parallel training (the reference wraps its nets in DistributedDataParallel: octfusion_model_union.py:185-196,
    octfusion_model_vae.py:121-130): average the gradient dict over the ranks in place.  Gradients are packed, in
    sorted key order, into flat fp32 buckets of up to `bucket_bytes` -- a few large ring all-reduces (per-link bound on
    xGMI, so size matters more than count) instead of one per tensor.  Every rank must hold the same keys; a key that
    is missing on some rank (a parameter without gradient there) is an error the caller has to resolve.
    Returns the number of bytes reduced."""
    if not _active(force):
        return 0
    world = dist.get_world_size()
    keys = sorted(grads)
    _check_gradient_set(grads, keys)
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


def max_over_ranks(value, device, force=False):
    """MAX all-reduce of a python float (timing)."""
    if not _active(force):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device, force=False):
    """[value of rank 0, ..., value of rank W-1] on every rank (per-rank step times)."""
    if not _active(force):
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def barrier(force=False):
    if _active(force):
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def shutdown():
    """Destroy the process group (tests that create a forced one-rank group)."""
    global _FORCED
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    _FORCED = False
