"""Oracle: octree container + Morton key codec (torch CPU).  TEST INFRASTRUCTURE.

Restates the slice of the third-party ``ocnn`` package (ocnn-pytorch, unpinned,
reference requirements.txt:1 -- NOT present under /root/reference) that the
reference hot path touches.  Semantics follow ocnn-pytorch's public API as
constrained by the reference's own call sites:

* ``keyd = key | depth << 58`` and ``key >> 48`` is the batch id
  (reference models/networks/dualoctree_networks/dual_octree.py:46,75);
* "for full octree, the octree key is the index" (dual_octree.py:137);
* child-octant bit order x->4, y->2, z->1 (dual_octree.py:85-94);
* ``children < 0`` <=> leaf (dual_octree.py:72,203-204);
* ``octree_grow`` does not bump ``depth`` (callers do ``octree.depth += 1``:
  utils/util_dualoctree.py:239-240, graph_vae.py:207-208);
* ``search_key(key, depth)`` returns the node index or -1 (mpu.py:72,78).

Parity at this boundary is UNPINNED (no reference tests, dependency absent).
"""
import torch

_BATCH_SHIFT = 48


def cumsum(data, dim, exclusive=False):
    """ocnn.utils.cumsum (call site dual_octree.py:30): exclusive => length n+1."""
    out = torch.cumsum(data, dim)
    if exclusive:
        size = list(data.size())
        size[dim] = 1
        out = torch.cat([torch.zeros(size, dtype=out.dtype, device=out.device), out], dim)
    return out


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    """ocnn.utils.scatter_add == reference .../utils/scatter.py:24-39."""
    if dim < 0:
        dim = src.dim() + dim
    idx = index
    if idx.dim() == 1:
        for _ in range(dim):
            idx = idx.unsqueeze(0)
    for _ in range(idx.dim(), src.dim()):
        idx = idx.unsqueeze(-1)
    idx = idx.expand_as(src)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif idx.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(idx.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, idx, src)


def xyz2key(x, y, z, b=None, depth=16):
    """Morton interleave: level-bit i of x -> key bit 3i+2, y -> 3i+1, z -> 3i."""
    x = x.to(torch.int64)
    y = y.to(torch.int64)
    z = z.to(torch.int64)
    key = torch.zeros_like(x)
    for i in range(depth):
        m = 1 << i
        key = key | ((x & m) << (2 * i + 2)) | ((y & m) << (2 * i + 1)) | ((z & m) << (2 * i))
    if b is not None:
        if torch.is_tensor(b):
            b = b.to(torch.int64)
        key = key | (b << _BATCH_SHIFT)
    return key


def key2xyz(key, depth=16):
    """Inverse of :func:`xyz2key`; returns (x, y, z, b) as int64."""
    key = key.to(torch.int64)
    b = key >> _BATCH_SHIFT
    k = key & ((1 << _BATCH_SHIFT) - 1)
    x = torch.zeros_like(k)
    y = torch.zeros_like(k)
    z = torch.zeros_like(k)
    for i in range(depth):
        x = x | ((k >> (3 * i + 2)) & 1) << i
        y = y | ((k >> (3 * i + 1)) & 1) << i
        z = z | ((k >> (3 * i)) & 1) << i
    return x, y, z, b


class Octree:
    """Batched octree: per-depth sorted keys (batch-major) + child pointers."""

    def __init__(self, depth, full_depth=2, batch_size=1, device='cpu', **kwargs):
        self.depth = depth
        self.full_depth = full_depth
        self.batch_size = batch_size
        self.device = device
        n = depth + 1
        self.keys = [None] * n
        self.children = [None] * n
        self.nnum = torch.zeros(n, dtype=torch.int64)
        self.nnum_nempty = torch.zeros(n, dtype=torch.int64)

    # -- queries ---------------------------------------------------------
    def key(self, depth, nempty=False):
        key = self.keys[depth]
        if nempty:
            key = key[self.nempty_mask(depth)]
        return key

    def xyzb(self, depth, nempty=False):
        return key2xyz(self.key(depth, nempty), depth)

    def batch_id(self, depth, nempty=False):
        return self.key(depth, nempty) >> _BATCH_SHIFT

    def nempty_mask(self, depth):
        return self.children[depth] >= 0

    def search_key(self, query, depth, nempty=False):
        """ocnn Octree.search_key (call site mpu.py:72): index of each query key in the sorted key list of
        `depth`, -1 when absent.  Published algorithm (ocnn-pytorch octree.py): searchsorted + equality test."""
        key = self.key(depth, nempty)
        query = query.to(torch.int64)
        idx = torch.searchsorted(key, query)
        inside = idx < key.shape[0]
        found = torch.zeros_like(inside)
        found[inside] = key[idx[inside]] == query[inside]
        return torch.where(found, idx, torch.full_like(idx, -1))

    # -- construction ----------------------------------------------------
    def octree_grow_full(self, depth, update_neigh=False):
        num = 8 ** depth
        self.nnum[depth] = num * self.batch_size
        self.nnum_nempty[depth] = num * self.batch_size
        key = torch.arange(num, dtype=torch.int64, device=self.device)
        bs = torch.arange(self.batch_size, dtype=torch.int64, device=self.device)
        self.keys[depth] = (key.unsqueeze(0) | (bs.unsqueeze(1) << _BATCH_SHIFT)).reshape(-1)
        self.children[depth] = torch.arange(
            num * self.batch_size, dtype=torch.int32, device=self.device)

    def octree_split(self, split, depth):
        split = split.to(torch.int64)
        children = torch.cumsum(split, dim=0) - 1
        children = torch.where(split > 0, children, torch.full_like(children, -1))
        self.children[depth] = children.to(torch.int32)
        self.nnum_nempty[depth] = int(split.sum())

    def octree_grow(self, depth, update_neigh=False):
        while len(self.keys) <= depth:      # defensive; ctor depth normally covers it
            self.keys.append(None)
            self.children.append(None)
            self.nnum = torch.cat([self.nnum, torch.zeros(1, dtype=torch.int64)])
            self.nnum_nempty = torch.cat([self.nnum_nempty, torch.zeros(1, dtype=torch.int64)])
        nnum = int(self.nnum_nempty[depth - 1]) * 8
        self.nnum[depth] = nnum
        self.nnum_nempty[depth] = nnum
        parent = self.key(depth - 1, nempty=True)
        b = parent >> _BATCH_SHIFT
        k = parent & ((1 << _BATCH_SHIFT) - 1)
        oct8 = torch.arange(8, dtype=torch.int64, device=self.device)
        key = ((k.unsqueeze(1) << 3) | oct8.unsqueeze(0)) | (b.unsqueeze(1) << _BATCH_SHIFT)
        self.keys[depth] = key.reshape(-1)
        self.children[depth] = torch.arange(nnum, dtype=torch.int32, device=self.device)

    def to(self, device):
        return self

    def cuda(self):
        return self


def octree2voxel(data, octree, depth, nempty=False):
    """ocnn.nn.octree2voxel (call site graph_unet_lr.py:176): channel-last dense grid."""
    x, y, z, b = octree.xyzb(depth, nempty)
    num = 1 << depth
    vox = data.new_zeros([octree.batch_size, num, num, num, data.shape[1]])
    vox[b, x, y, z] = data
    return vox


def octree_pad(data, octree, depth, val=0.0):
    """ocnn.nn.octree_pad (call site util_dualoctree.py:204): nempty rows -> all rows."""
    mask = octree.nempty_mask(depth)
    out = data.new_full([int(octree.nnum[depth]), data.shape[1]], val)
    out[mask] = data
    return out
