"""Oracle: NeuralMPU SDF evaluation (torch CPU).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates reference models/networks/dualoctree_networks/mpu.py:55-153 together with the two sparse products
it calls (utils/spmm.py:12-61) as one dense-per-point computation: for every query point, every depth d in
[depth_start, depth_end] and each of the 8 cell centres around the point at that depth,

    w   = prod_axis(1 - |f|) * d**2 / 50            (mpu.py:86-91; f = offset to the centre in cells)
    val = code[row] . [f * 2 / 2**d, 1]              (spmm.py:54-57; mpu.py:95)
    sdf = sum(w * val) / (sum(w) + 1e-8)             (mpu.py:127-129)

over the centres that exist in the octree (mpu.py:72-78) and, below depth_end, are leaves (mpu.py:113-116);
row = node index + (nodes of depths depth_start .. d-1) (mpu.py:118).  mask = any centre of depth_end exists
(mpu.py:132).  Pinned against the reference's own mpu.py through tests/golden/g_mpu.pt.
"""
import torch

from .octree import xyz2key


def _corners():
    # mpu.py:37-52: (0,0,0), (0,0,1), (0,1,0), ... x slowest
    return torch.tensor([[i >> 2 & 1, i >> 1 & 1, i & 1] for i in range(8)], dtype=torch.float32)


def mpu_abs(f):
    """mpu.py:18-32 (class ABS): |f| whose derivative at 0 is +1 (torch.abs gives 0 there)."""
    return torch.where(f < 0, -f, f)


def linear_pts(octree, depth, pts):
    """mpu.py:55-96 for one depth; returns per (point, corner): node index (-1 absent/out of bounds),
    offsets in cells, weight."""
    scale = 2 ** depth
    xyz = (pts[:, :3] + 1.0) * (scale / 2.0) - 0.5
    base = torch.floor(xyz).detach()                                   # mpu.py:63
    corners = base.unsqueeze(1) + _corners().to(pts.device)            # [n, 8, 3]
    f = xyz.unsqueeze(1) - corners                                     # [n, 8, 3]
    b = pts[:, 3:4].expand(-1, 8)
    c16 = corners.to(torch.int16)                                      # mpu.py:70 goes through .short()
    key = xyz2key(c16[..., 0].reshape(-1), c16[..., 1].reshape(-1), c16[..., 2].reshape(-1),
                  b.reshape(-1).to(torch.int16))
    idx = octree.search_key(key, depth).view(-1, 8)
    inb = ((corners > -1) & (corners < scale)).all(-1)
    idx = torch.where(inb, idx, torch.full_like(idx, -1))
    w = (1.0 - mpu_abs(f)).prod(-1) * (depth ** 2 / 50)
    return idx, f, w


def linear_pred(pts, octree, code, depth_start, depth_end):
    """mpu.py:97-134: (sdf [n], mask [n])."""
    n = pts.shape[0]
    num = torch.zeros(n, dtype=torch.float32)
    den = torch.zeros(n, dtype=torch.float32)
    base = 0
    mask = None
    for d in range(depth_start, depth_end + 1):
        idx, f, w = linear_pts(octree, d, pts)
        ok = idx >= 0
        if d == depth_end:
            mask = ok.any(-1)
        else:
            leaf = torch.zeros_like(ok)
            leaf[ok] = octree.children[d][idx[ok]] < 0
            ok = leaf
        rows = (idx + base).clamp(min=0)
        c = code[rows.view(-1)].view(n, 8, 4)
        val = (c[..., :3] * (f * (2.0 / 2 ** d))).sum(-1) + c[..., 3]
        wz = torch.where(ok, w, torch.zeros_like(w))
        num = num + (wz * val).sum(-1)
        den = den + wz.sum(-1)
        base += int(octree.nnum[d])
    return num / (den + 1e-8), mask


def neural_mpu(pos, reg_voxs, octree, full_depth, depth_stop, depth):
    """mpu.py:137-153: {d: (sdf, mask)} for d in depth_stop..depth."""
    return {d: linear_pred(pos, octree, reg_voxs[d], full_depth, d) for d in range(depth_stop, depth + 1)}
