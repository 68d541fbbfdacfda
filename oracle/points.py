"""Oracle: point cloud -> octree (torch CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates the third-party ocnn-pytorch routines behind the reference call sites
models/octfusion_model_union.py:198-212 / models/octfusion_model_vae.py:133-141 (`Octree.build_octree`,
`merge_octrees`) and dual_octree.py:343-360 (`InputFeature('ND')`).  ocnn is absent from /root/reference
(requirements.txt:1, unpinned), so this follows its PUBLISHED algorithm (ocnn/octree/octree.py: key every point,
unique-sort, then bottom-up `key >> 3` + unique_consecutive per depth; merge by batch bits + child offsets) --
PARITY UNPINNED at this boundary.  What IS pinned from the reference side: the octree it yields must survive
octree2split_small/large -> split2octree_small/large (utils/util_dualoctree.py:199-273, pinned by g_octree_graph)
unchanged, which tests/ checks.  Deliberately a different algorithm (bottom-up, unique-based, one octree per shape
then merge) from the product's batched top-down build, so agreement is a real cross-check.
"""
import torch
import torch.nn.functional as F

from .octree import Octree, scatter_add, xyz2key


def build_octree(points, normals, depth, full_depth):
    """ocnn Octree.build_octree for one shape (batch size 1).  Returns (octree, avg_points [nnum_nempty_depth, 3],
    avg_normals [.., 3]) -- ocnn's octree.points[depth] / octree.normals[depth]."""
    oc = Octree(depth, full_depth, 1)
    scale = 2 ** (depth - 1)
    pts = (points + 1.0) * scale
    hi = 2 ** depth - 1
    ijk = pts.long().clamp(0, hi)          # (clipped inputs lie strictly inside the cube: the clamp never acts on them)
    key = xyz2key(ijk[:, 0], ijk[:, 1], ijk[:, 2], None, depth)
    node_key, idx, counts = torch.unique(key, sorted=True, return_inverse=True, return_counts=True)
    for d in range(full_depth + 1):
        oc.octree_grow_full(d)
    for d in range(depth, full_depth, -1):
        pkey = node_key >> 3
        pkey, pidx = torch.unique_consecutive(pkey, return_inverse=True)
        key_d = (pkey.unsqueeze(-1) * 8 + torch.arange(8)).view(-1)
        oc.keys[d] = key_d
        oc.nnum[d] = key_d.numel()
        oc.nnum_nempty[d] = node_key.numel()
        addr = (pidx << 3) | (node_key % 8)
        children = -torch.ones(key_d.numel(), dtype=torch.int32)
        children[addr] = torch.arange(node_key.numel(), dtype=torch.int32)
        oc.children[d] = children
        node_key = pkey
    children = -torch.ones_like(oc.children[full_depth])
    children[node_key] = torch.arange(node_key.numel(), dtype=torch.int32)
    oc.children[full_depth] = children
    oc.nnum_nempty[full_depth] = node_key.numel()
    avg_pts = scatter_add(pts, idx, dim=0) / counts.unsqueeze(1)
    avg_nrm = F.normalize(scatter_add(normals, idx, dim=0)) if normals is not None else None
    return oc, avg_pts, avg_nrm


def merge_octrees(octrees):
    """ocnn.octree.merge_octrees: batch id into key bits 48.., child pointers offset by the preceding elements'
    non-empty counts."""
    first = octrees[0]
    out = Octree(first.depth, first.full_depth, len(octrees))
    for d in range(first.depth + 1):
        keys, children, off = [], [], 0
        for i, oc in enumerate(octrees):
            keys.append(oc.keys[d] | (i << 48))
            c = oc.children[d].clone()
            c[c >= 0] += off
            children.append(c)
            off += int(oc.nnum_nempty[d])
        out.keys[d], out.children[d] = torch.cat(keys), torch.cat(children)
        out.nnum[d] = sum(int(oc.nnum[d]) for oc in octrees)
        out.nnum_nempty[d] = off
    return out


def input_feature_nd(octree, avg_pts, avg_nrm):
    """ocnn Octree.get_input_feature('ND', nempty=False): [nnum_depth, 4], zero rows for empty nodes."""
    local = avg_pts.frac() - 0.5
    dis = (local * avg_nrm).sum(dim=1, keepdim=True)
    feat = torch.cat([avg_nrm, dis], dim=1)
    out = torch.zeros(int(octree.nnum[octree.depth]), 4)
    out[octree.nempty_mask(octree.depth)] = feat
    return out


def points2octree_batch(points_list, normals_list, depth, full_depth):
    """octfusion_model_union.py:199-209: points2octree per shape, merge_octrees; plus the merged 'ND' feature."""
    built = [build_octree(p, n, depth, full_depth) for p, n in zip(points_list, normals_list)]
    merged = merge_octrees([b[0] for b in built])
    feat = torch.cat([input_feature_nd(b[0], b[1], b[2]) for b in built])
    return merged, feat
