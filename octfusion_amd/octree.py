"""Device-resident octree container + split<->octree conversions.

Mirrors the slice of ocnn.octree.Octree the reference touches on the sampling
path with the same attribute / method names, and the reference's
``create_full_octree`` (ldm_diffusion_util.py:318-325), ``split2octree_small`` /
``split2octree_large`` (utils/util_dualoctree.py:225-273).  All per-node work
(label extraction, split = exclusive scan, grow = key expansion) runs in
libofx HIP kernels; only the per-depth node COUNTS come back to the host (they
size the next allocation).
"""
import copy

import torch

from . import _lib
from ._lib import call, ptr, stream


class Octree:
    def __init__(self, depth, full_depth=2, batch_size=1, device='cuda', **kwargs):
        _lib.require_device()
        self.depth = depth
        self.full_depth = full_depth
        self.batch_size = batch_size
        self.device = torch.device(device)
        n = depth + 1
        self.keys = [None] * n
        self.children = [None] * n
        self.nnum = torch.zeros(n, dtype=torch.int64)          # host
        self.nnum_nempty = torch.zeros(n, dtype=torch.int64)   # host

    # ---- queries (ocnn API names) -------------------------------------
    def nempty_mask(self, depth):
        return self.children[depth] >= 0

    def key(self, depth, nempty=False):
        k = self.keys[depth]
        return k[self.nempty_mask(depth)] if nempty else k

    def batch_id(self, depth, nempty=False):
        return self.key(depth, nempty) >> 48

    def xyzb(self, depth, nempty=False):
        # index plumbing for callers that want coordinates; the kernels decode keys themselves
        k = self.key(depth, nempty)
        b = k >> 48
        k = k & ((1 << 48) - 1)
        x = torch.zeros_like(k)
        y = torch.zeros_like(k)
        z = torch.zeros_like(k)
        for i in range(depth):
            x |= ((k >> (3 * i + 2)) & 1) << i
            y |= ((k >> (3 * i + 1)) & 1) << i
            z |= ((k >> (3 * i)) & 1) << i
        return x, y, z, b

    # ---- construction --------------------------------------------------
    def octree_grow_full(self, depth, update_neigh=False):
        n = (8 ** depth) * self.batch_size
        self.keys[depth] = torch.empty(n, dtype=torch.int64, device=self.device)
        self.children[depth] = torch.empty(n, dtype=torch.int32, device=self.device)
        call('ofx_octree_full_layer', depth, self.batch_size, ptr(self.keys[depth]),
             ptr(self.children[depth]), stream())
        self.nnum[depth] = n
        self.nnum_nempty[depth] = n

    def octree_split(self, split, depth):
        label = split.to(device=self.device, dtype=torch.int32).contiguous()
        n = label.numel()
        if n != int(self.nnum[depth]):
            raise ValueError('octree_split: %d labels for %d nodes' % (n, int(self.nnum[depth])))
        children = torch.empty(n, dtype=torch.int32, device=self.device)
        scan = torch.empty(n + 1, dtype=torch.int32, device=self.device)
        ws = torch.empty(_lib.lib().ofx_scan_ws_bytes(n), dtype=torch.uint8, device=self.device)
        call('ofx_octree_split', ptr(label), n, ptr(children), ptr(scan), ptr(ws), stream())
        self.children[depth] = children
        self.nnum_nempty[depth] = int(scan[n].item())        # sizes the next layer (host sync)

    def octree_grow(self, depth, update_neigh=False):
        while len(self.keys) <= depth:
            self.keys.append(None)
            self.children.append(None)
            self.nnum = torch.cat([self.nnum, torch.zeros(1, dtype=torch.int64)])
            self.nnum_nempty = torch.cat([self.nnum_nempty, torch.zeros(1, dtype=torch.int64)])
        n = int(self.nnum_nempty[depth - 1]) * 8
        self.keys[depth] = torch.empty(n, dtype=torch.int64, device=self.device)
        self.children[depth] = torch.empty(n, dtype=torch.int32, device=self.device)
        call('ofx_octree_grow', ptr(self.keys[depth - 1]), ptr(self.children[depth - 1]),
             int(self.nnum[depth - 1]), ptr(self.keys[depth]), ptr(self.children[depth]), stream())
        self.nnum[depth] = n
        self.nnum_nempty[depth] = n

    # ---- construction from a point cloud (ocnn Octree.build_octree; call sites
    #      models/octfusion_model_union.py:198-212, models/octfusion_model_vae.py:133-141) ------------------
    def build_octree(self, point_cloud):
        """Build the octree of `point_cloud` (a Points, optionally batched through `batch_id`) down to self.depth:
        one key sort for the whole batch, then per depth a binary search per node decides which nodes split.
        Also computes the 'ND' input feature of the finest depth (get_input_feature)."""
        pts = point_cloud.points
        if pts.dtype != torch.float32 or not pts.is_cuda:
            raise _lib.OfxError('build_octree needs fp32 HIP tensors (no CPU path)')
        pts = pts.contiguous()
        nrm = point_cloud.normals.contiguous() if point_cloud.normals is not None else None
        bid = point_cloud.batch_id
        if bid is not None:
            bid = bid.reshape(-1).to(torch.int32).contiguous()
        self.batch_size = int(point_cloud.batch_size)
        dev = self.device = pts.device
        n = pts.shape[0]
        depth = self.depth
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        idx = torch.empty(n, dtype=torch.int32, device=dev)
        call('ofx_points_keys', ptr(pts), pts.stride(0), ptr(bid) if bid is not None else None, 0, n, depth, ptr(keys),
             ptr(idx), stream())
        skeys, sidx = torch.empty_like(keys), torch.empty_like(idx)
        nbytes = _lib.lib().ofx_points_sort_ws_bytes(n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        call('ofx_points_sort', ptr(keys), ptr(idx), n, ptr(skeys), ptr(sidx), ptr(ws), nbytes, stream())
        for d in range(self.full_depth + 1):
            self.octree_grow_full(d)
        for d in range(self.full_depth, depth + 1):
            nd = int(self.nnum[d])
            label = torch.empty(nd, dtype=torch.int32, device=dev)
            call('ofx_octree_label_from_points', ptr(skeys), n, ptr(self.keys[d]), nd, depth, d, ptr(label), stream())
            self.octree_split(label, d)
            if d < depth:
                self.octree_grow(d + 1)
        nd = int(self.nnum[depth])
        feat = torch.empty(nd, 4, dtype=torch.float32, device=dev)
        avg = torch.empty(nd, 3, dtype=torch.float32, device=dev)
        call('ofx_octree_point_features', ptr(skeys), ptr(sidx), n, ptr(pts), pts.stride(0),
             ptr(nrm) if nrm is not None else None, nrm.stride(0) if nrm is not None else 0, ptr(self.keys[depth]), nd,
             depth, ptr(feat), ptr(avg), None, stream())
        self._feature_nd = feat
        self._avg_points = avg
        self._has_normals = nrm is not None
        return self

    def get_input_feature(self, feature='ND', nempty=False):
        """ocnn Octree.get_input_feature for the signals the reference asks for (dual_octree.py:345: 'ND'):
        N = averaged unit normal, D = dot(frac(mean position) - 0.5, normal); rows of empty nodes are zero
        (octree_pad) unless nempty."""
        feat = getattr(self, '_feature_nd', None)
        if feat is None:
            raise _lib.OfxError('get_input_feature: the octree was not built from a point cloud')
        cols = []
        for ch in feature.upper():
            if ch == 'N':
                cols.append(feat[:, :3])
            elif ch == 'D':
                cols.append(feat[:, 3:4])
            else:
                raise ValueError('unsupported feature %r (the reference uses "ND")' % ch)
        out = cols[0] if len(cols) == 1 else torch.cat(cols, dim=1)
        return out[self.nempty_mask(self.depth)] if nempty else out.contiguous()

    def to(self, device):
        return self

    def cuda(self):
        return self

    def batch_slices(self, bounds):
        """Sub-octrees of consecutive batch ranges [bounds[i], bounds[i+1]): the inverse of merge_octrees for runs of
        elements.  Nodes of a depth are sorted by key with the batch id on top (bits 48..), so every element range is
        one contiguous run per depth: keys lose the range's first batch id, child pointers the non-empty nodes in front
        of the run.  Index plumbing on the per-depth arrays; two host reads for all ranges together (run bounds, then
        non-empty counts -- they size the sub-octrees' host tables).  Used by sampler.sample_loop to run a batch as
        independent lanes on separate HIP streams."""
        bounds = [int(b) for b in bounds]
        if bounds[0] < 0 or bounds[-1] > self.batch_size or any(a >= b for a, b in zip(bounds, bounds[1:])):
            raise ValueError('batch_slices: bounds %r for batch size %d' % (bounds, self.batch_size))
        nd = self.depth + 1
        edges = torch.tensor(bounds, dtype=torch.int64, device=self.device) << 48
        cut = torch.stack([torch.searchsorted(self.keys[d], edges) for d in range(nd)]).tolist()     # host read 1
        csum = [torch.cumsum((self.children[d] >= 0).to(torch.int64), 0) for d in range(nd)]
        zero = torch.zeros(1, dtype=torch.int64, device=self.device)
        before = torch.stack([torch.cat([zero, csum[d]])[torch.tensor(cut[d], device=self.device)]
                              for d in range(nd)]).tolist()                                           # host read 2
        out = []
        for i, (b0, b1) in enumerate(zip(bounds, bounds[1:])):
            oc = Octree(self.depth, self.full_depth, b1 - b0, self.device)
            oc.keys, oc.children = [None] * nd, [None] * nd
            oc.nnum, oc.nnum_nempty = torch.zeros(nd, dtype=torch.int64), torch.zeros(nd, dtype=torch.int64)
            for d in range(nd):
                lo, hi = cut[d][i], cut[d][i + 1]
                oc.keys[d] = self.keys[d][lo:hi] - (b0 << 48)
                c = self.children[d][lo:hi]
                off = before[d][i]
                oc.children[d] = torch.where(c >= 0, c - off, c) if off else c.clone()
                oc.nnum[d] = hi - lo
                oc.nnum_nempty[d] = before[d][i + 1] - off
            feat = getattr(self, '_feature_nd', None)
            if feat is not None:
                lo, hi = cut[self.depth][i], cut[self.depth][i + 1]
                oc._feature_nd = feat[lo:hi]
                oc._avg_points = self._avg_points[lo:hi]
            out.append(oc)
        return out


class Points:
    """ocnn.octree.Points as the reference uses it (datasets/dualoctree_snet.py:39-47): positions in [-1, 1],
    optional normals / features, optional per-point batch id."""

    def __init__(self, points, normals=None, features=None, labels=None, batch_id=None, batch_size=1):
        self.points, self.normals, self.features, self.labels = points, normals, features, labels
        self.batch_id, self.batch_size = batch_id, batch_size
        self.device = points.device

    def inbox_mask(self, min=-1.0, max=1.0):
        """STRICTLY inside the box, as ocnn-pytorch's Points.inbox_mask (points > bbmin and points < bbmax; ocnn is
        absent here -- parity at the ocnn boundary is unpinned, SURVEY 8c -- and this is its published behaviour): a
        point ON a face of the cube is dropped, so no point ever lands outside the [0, 2^depth) cell range and the
        clamp in ofx_points_keys never moves a point the reference's data path would have kept."""
        return ((self.points > min) & (self.points < max)).all(dim=1)

    def clip(self, min=-1.0, max=1.0):
        """keep the points strictly inside (min, max)^3; returns the mask (index plumbing: one boolean gather per array)."""
        mask = self.inbox_mask(min, max)
        for name in ('points', 'normals', 'features', 'labels', 'batch_id'):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, v[mask])
        return mask

    def cuda(self, non_blocking=False):
        for name in ('points', 'normals', 'features', 'labels', 'batch_id'):
            v = getattr(self, name)
            if v is not None:
                setattr(self, name, v.cuda(non_blocking=non_blocking))
        self.device = self.points.device
        return self

    to = lambda self, device, **k: self.cuda() if torch.device(device).type == 'cuda' else self   # noqa: E731


def merge_points(points_list):
    """ocnn.octree.merge_points: one batched Points with batch_id = position in the list."""
    cat = lambda xs: torch.cat(xs, dim=0) if all(x is not None for x in xs) else None      # noqa: E731
    bid = torch.cat([torch.full((p.points.shape[0],), i, dtype=torch.int32, device=p.points.device)
                     for i, p in enumerate(points_list)])
    return Points(cat([p.points for p in points_list]), cat([p.normals for p in points_list]),
                  cat([p.features for p in points_list]), None, bid, len(points_list))


def build_octree_batch(points_list, depth, full_depth):
    """points2octree per shape + merge_octrees (octfusion_model_union.py:199-209) as ONE build: the batch is keyed,
    sorted and grown together, so there are no per-shape octrees to merge."""
    pts = merge_points(points_list)
    return Octree(depth, full_depth, len(points_list), pts.points.device).build_octree(pts)


def merge_octrees(octrees):
    """ocnn.octree.merge_octrees for octrees built separately (batch size 1 each): per depth, keys get the element's
    batch id in bits 48.. and child pointers are shifted by the non-empty nodes of the elements before it.
    (Index plumbing on the per-depth arrays; build_octree_batch avoids it altogether.)"""
    first = octrees[0]
    depth, fd = first.depth, first.full_depth
    out = Octree(depth, fd, len(octrees), first.device)
    for d in range(depth + 1):
        keys, children = [], []
        off = 0
        for i, oc in enumerate(octrees):
            assert oc.batch_size == 1 and oc.depth == depth and oc.full_depth == fd
            keys.append((oc.keys[d] & ((1 << 48) - 1)) | (i << 48))
            c = oc.children[d]
            children.append(torch.where(c >= 0, c + off, c))
            off += int(oc.nnum_nempty[d])
        out.keys[d] = torch.cat(keys)
        out.children[d] = torch.cat(children)
        out.nnum[d] = sum(int(oc.nnum[d]) for oc in octrees)
        out.nnum_nempty[d] = off
    feats = [getattr(oc, '_feature_nd', None) for oc in octrees]
    if all(f is not None for f in feats):
        out._feature_nd = torch.cat(feats)
        out._avg_points = torch.cat([oc._avg_points for oc in octrees])
    return out


def create_full_octree(depth, full_depth, batch_size, device='cuda'):
    """ldm_diffusion_util.py:318-325."""
    octree = Octree(depth, full_depth, batch_size, device)
    for d in range(full_depth + 1):
        octree.octree_grow_full(d)
    octree.depth = full_depth
    return octree


def split2octree_small(split, input_depth, full_depth):
    """utils/util_dualoctree.py:225-250: [B, 8, S, S, S] split codes -> octree of depth full_depth+2."""
    if not split.is_cuda:
        raise _lib.OfxError('split2octree_small needs a HIP tensor (no CPU path)')
    split = split.contiguous().float()
    B = split.shape[0]
    S = 1 << full_depth
    if tuple(split.shape) != (B, 8, S, S, S):
        raise ValueError('split must be [B, 8, %d, %d, %d]' % (S, S, S))
    octree = create_full_octree(input_depth, full_depth, B, split.device)
    n0 = int(octree.nnum[full_depth])
    label0 = torch.empty(n0, dtype=torch.int32, device=split.device)
    call('ofx_split_small_label0', ptr(split), B, full_depth, ptr(label0), stream())
    octree.octree_split(label0, full_depth)
    octree.octree_grow(full_depth + 1)
    octree.depth += 1
    label1 = torch.empty(int(octree.nnum[full_depth + 1]), dtype=torch.int32, device=split.device)
    call('ofx_split_small_label1', ptr(split), B, full_depth, ptr(octree.children[full_depth]),
         ptr(label1), stream())
    octree.octree_split(label1, full_depth + 1)
    octree.octree_grow(full_depth + 2)
    octree.depth += 1
    return octree


def split2octree_large(octree, split, small_depth):
    """utils/util_dualoctree.py:252-273: [nnum[small_depth], 8] split codes -> two more levels."""
    if not split.is_cuda:
        raise _lib.OfxError('split2octree_large needs a HIP tensor (no CPU path)')
    split = split.contiguous().float()
    n = int(octree.nnum[small_depth])
    if tuple(split.shape) != (n, 8):
        raise ValueError('split must be [%d, 8]' % n)
    out = copy.copy(octree)
    out.keys = list(octree.keys)
    out.children = list(octree.children)
    out.nnum = octree.nnum.clone()
    out.nnum_nempty = octree.nnum_nempty.clone()
    label0 = torch.empty(n, dtype=torch.int32, device=split.device)
    call('ofx_split_large_label0', ptr(split), n, ptr(label0), stream())
    out.octree_split(label0, small_depth)
    out.octree_grow(small_depth + 1)
    out.depth += 1
    label1 = torch.empty(int(out.nnum[small_depth + 1]), dtype=torch.int32, device=split.device)
    call('ofx_split_large_label1', ptr(split), n, ptr(out.children[small_depth]), ptr(label1), stream())
    out.octree_split(label1, small_depth + 1)
    out.octree_grow(small_depth + 2)
    out.depth += 1
    return out


def _child_occupancy(octree, depth):
    """[nnum[depth], 8] fp32 in {0, 1}: child j of node i (at depth+1) is non-empty; rows of empty nodes are 0
    (ocnn.nn.octree_pad of the reshaped nempty mask, util_dualoctree.py:201-204 / 215-218)."""
    nz = (octree.children[depth + 1] >= 0).reshape(-1, 8).float()
    out = torch.zeros(int(octree.nnum[depth]), 8, dtype=torch.float32, device=nz.device)
    out[octree.nempty_mask(depth)] = nz
    return out


def octree2split_small(octree, full_depth):
    """utils/util_dualoctree.py:199-211 (inverse of split2octree_small; the diffusion stage-1 data format):
    -> [B, 8, S, S, S] in {-1, +1}."""
    from . import ops
    occ = _child_occupancy(octree, full_depth)
    return 2 * ops.octree2voxel_cf(occ, octree.batch_size, full_depth) - 1


def octree2split_large(octree, small_depth):
    """utils/util_dualoctree.py:213-223 (inverse of split2octree_large; the stage-2 data format):
    -> [nnum[small_depth], 8] in {-1, +1}."""
    return 2 * _child_occupancy(octree, small_depth) - 1
