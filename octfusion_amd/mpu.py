"""NeuralMPU: SDF values at query points from the per-node 4-vectors the GraphVAE decoder regresses.

Host-side mirror of reference models/networks/dualoctree_networks/mpu.py:137-153 (`NeuralMPU`) and of the
sampling sweep utils/util_dualoctree.py:99-118 (`calc_sdf`); all arithmetic runs in libofx
(`ofx_mpu_eval` / `ofx_mpu_eval_grid`, csrc/ofx_graph.hip).  No CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import call, ptr, stream


class _TreeHandle:
    """Depth-concatenated child / key arrays of an Octree + the ofx_tree_t that points at them."""

    def __init__(self, octree):
        depth = octree.depth
        self.child = torch.cat([octree.children[d] for d in range(depth + 1)]).contiguous()
        self.key = torch.cat([octree.keys[d] for d in range(depth + 1)]).contiguous()
        self.nnum = [int(v) for v in octree.nnum[:depth + 1]]
        self.sig = _TreeHandle.signature(octree)
        self._nnum_c = (ctypes.c_int64 * (depth + 1))(*self.nnum)
        self._nne_c = (ctypes.c_int64 * (depth + 1))(*[int(v) for v in octree.nnum_nempty[:depth + 1]])
        self.tree = _lib.OfxTree(depth, octree.full_depth, octree.batch_size, ptr(self.child), ptr(self.key), None,
                                 ctypes.addressof(self._nnum_c), ctypes.addressof(self._nne_c))

    @staticmethod
    def signature(octree):
        """identity of every per-depth array: octree_split / octree_grow REPLACE children[d] / keys[d] (also when the
        node count does not change, e.g. the final split at depth_out), which must invalidate the handle"""
        return tuple((octree.children[d].data_ptr(), octree.children[d]._version, octree.keys[d].data_ptr(),
                      int(octree.nnum[d])) for d in range(octree.depth + 1))

    @staticmethod
    def of(octree):
        h = getattr(octree, '_ofx_mpu_tree', None)
        if h is None or h.sig != _TreeHandle.signature(octree):
            h = _TreeHandle(octree)
            octree._ofx_mpu_tree = h
        return h


def _code(reg, tree, depth_start, depth_end):
    rows = sum(tree.nnum[depth_start:depth_end + 1])
    if reg.dtype != torch.float32 or not reg.is_cuda:
        raise _lib.OfxError('NeuralMPU needs fp32 HIP tensors (no CPU path)')
    assert reg.shape == (rows, 4), 'reg_voxs has shape %s, expected (%d, 4)' % (tuple(reg.shape), rows)
    return reg.contiguous()


def mpu_eval(octree, depth_start, depth_end, pos, reg):
    """(sdf [n], mask [n] bool) -- get_linear_pred (mpu.py:97-134) for one target depth."""
    h = _TreeHandle.of(octree)
    reg = _code(reg, h, depth_start, depth_end)
    if pos.dtype != torch.float32 or not pos.is_cuda:
        raise _lib.OfxError('NeuralMPU needs fp32 HIP tensors (no CPU path)')
    pos = pos.contiguous()
    n = pos.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=pos.device)
    mask = torch.empty(n, dtype=torch.uint8, device=pos.device)
    call('ofx_mpu_eval', ctypes.byref(h.tree), depth_start, depth_end, ptr(pos), n, ptr(reg), ptr(sdf), ptr(mask),
         stream())
    return sdf, mask.bool()


class NeuralMPU:
    """mpu.py:137-153.  __call__(pos [n,4], reg_voxs {d: [rows_d,4]}, octree) -> {d: (sdf [n], mask [n])}."""

    def __init__(self, full_depth, depth_stop, depth):
        self.full_depth = full_depth
        self.depth_stop = depth_stop
        self.depth = depth

    def __call__(self, pos, reg_voxs, octree_out):
        return {d: mpu_eval(octree_out, self.full_depth, d, pos, reg_voxs[d])
                for d in range(self.depth_stop, self.depth + 1)}


class MpuField:
    """The `_neural_mpu(pos)` closure of GraphVAE.decode_code (graph_vae.py:319-323) as an object, so the SDF
    sweep can also ask for lattice points to be generated inside the kernel."""

    def __init__(self, full_depth, depth_out, reg, octree):
        self.full_depth = full_depth
        self.depth_out = depth_out
        self.reg = reg
        self.octree = octree

    def __call__(self, pos):
        return mpu_eval(self.octree, self.full_depth, self.depth_out, pos, self.reg)[0]

    def grid(self, size, bbmin, bbmax, batch_index, head, count, out):
        h = _TreeHandle.of(self.octree)
        reg = _code(self.reg, h, self.full_depth, self.depth_out)
        step = (bbmax - bbmin) / size
        call('ofx_mpu_eval_grid', ctypes.byref(h.tree), self.full_depth, self.depth_out, ptr(reg), size, step, bbmin,
             batch_index, head, count, ptr(out), None, stream())


def calc_sdf(model, batch_size=1, size=256, max_batch=64 ** 3, bbmin=-1.0, bbmax=1.0):
    """utils/util_dualoctree.py:99-118: SDF on the size^3 lattice fl(i * (bbmax-bbmin)/size + bbmin) for every
    batch element -> [batch_size, size, size, size] (x slowest).  `model` is an MpuField (lattice points are
    generated in the kernel: nothing but the 4 B/point result touches HBM) or any callable pts[n,4] -> sdf[n]."""
    num = size ** 3
    dev = model.reg.device if isinstance(model, MpuField) else torch.device('cuda')
    sdfs = torch.empty(batch_size, num, dtype=torch.float32, device=dev)
    if isinstance(model, MpuField):
        for b in range(batch_size):
            model.grid(size, float(bbmin), float(bbmax), b, 0, num, sdfs[b])
        return sdfs.view(batch_size, size, size, size)
    step = torch.tensor((bbmax - bbmin) / size, dtype=torch.float32, device=dev)
    lo = torch.tensor(bbmin, dtype=torch.float32, device=dev)
    for b in range(batch_size):
        head = 0
        while head < num:
            tail = min(head + max_batch, num)
            q = torch.arange(head, tail, device=dev)
            ijk = torch.stack([q // (size * size), (q // size) % size, q % size], 1).float()
            pts = torch.cat([ijk * step + lo, torch.full((tail - head, 1), float(b), device=dev)], 1)
            sdfs[b, head:tail] = model(pts).reshape(-1)
            head = tail
    return sdfs.view(batch_size, size, size, size)
