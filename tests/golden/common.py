"""Shared, deterministic input builders for golden generation and the tests.

Pure torch (CPU); no reference, oracle or product imports.
"""
import hashlib
import zlib

import torch


def fill_param(key, shape):
    """Deterministic, non-degenerate parameter for `key` (zero-init modules included)."""
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    t = torch.randn(shape, generator=g)
    eff = [s for s in shape if s != 1]
    if len(eff) <= 1:
        if key.endswith('weight') or key.endswith('weights'):
            if 'time_pos_emb' in key:
                return t
            return 1.0 + 0.1 * t
        return 0.1 * t
    numel = t.numel()
    if key.endswith('.weights') and len(shape) == 2:      # GraphConv [K, Cout]
        fan = numel // shape[1]
    else:
        fan = numel // shape[0]
    return t * (1.5 / max(1, fan) ** 0.5)


def rand_input(name, *shape):
    """Deterministic N(0,1) input tensor regenerated from its name (not stored in fixtures)."""
    g = torch.Generator().manual_seed(zlib.crc32(('in:' + name).encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g)


def fill_state_dict(keys_shapes):
    return {k: fill_param(k, s) for k, s in keys_shapes}


def sha_int(t):
    """sha256 of an integer tensor as little-endian int64 bytes."""
    return hashlib.sha256(t.to(torch.int64).contiguous().numpy().tobytes()).hexdigest()


def random_split_small(B, full_depth, seed, p=0.3):
    s = 1 << full_depth
    g = torch.Generator().manual_seed(seed)
    occ = (torch.rand(B, 8, s, s, s, generator=g) < p).float()
    # make some coarse cells completely empty so there are leaves at full_depth
    kill = (torch.rand(B, 1, s, s, s, generator=g) < 0.5).float()
    occ = occ * (1 - kill)
    return 2 * occ - 1


def random_split_large(nnum, seed, p=0.5):
    g = torch.Generator().manual_seed(seed)
    occ = (torch.rand(nnum, 8, generator=g) < p).float()
    kill = (torch.rand(nnum, 1, generator=g) < 0.3).float()
    return 2 * occ * (1 - kill) - 1


def shell6_split(B, jitter=False):
    """SURVEY.md section 8(d) 'shell-6': depth-5 occupancy 9 < |p-15.5| < 11 on 32^3."""
    g = torch.arange(32, dtype=torch.float32)
    X, Y, Z = torch.meshgrid(g, g, g, indexing='ij')
    r = torch.sqrt((X - 15.5) ** 2 + (Y - 15.5) ** 2 + (Z - 15.5) ** 2)
    out = []
    for b in range(B):
        dr = 0.25 * (b % 4) if jitter else 0.0
        occ = ((r > 9 + dr) & (r < 11 + dr)).float()
        s = torch.zeros(8, 16, 16, 16)
        for dx in range(2):
            for dy in range(2):
                for dz in range(2):
                    s[4 * dx + 2 * dy + dz] = 2 * occ[dx::2, dy::2, dz::2] - 1
        out.append(s)
    return torch.stack(out)


def shell8_split_large(x, y, z):
    """'shell-8': depth-7 occupancy 38.5 < |p-63.5| < 41.0 on 128^3, sampled per depth-6 node.

    x, y, z: int64 coordinates (64^3 grid) of ALL depth-6 nodes; returns [nnum6, 8] in {-1,+1}.
    """
    cols = []
    for dx in range(2):
        for dy in range(2):
            for dz in range(2):
                px = (2 * x + dx).float() - 63.5
                py = (2 * y + dy).float() - 63.5
                pz = (2 * z + dz).float() - 63.5
                r = torch.sqrt(px * px + py * py + pz * pz)
                cols.append(((r > 38.5) & (r < 41.0)).float())
    return 2 * torch.stack(cols, dim=1) - 1


TINY_HR_CFG = dict(image_size=16, input_depth=5, full_depth=3, in_channels=3, model_channels=32,
                   lr_model_channels=16, out_channels=3, num_res_blocks=[1, 1, 0],
                   channel_mult=[1, 2, 4], dims=3, num_classes=None, num_heads=4)
TINY_LR_CFG = dict(full_depth=3, in_split_channels=8, model_channels=16, out_split_channels=8,
                   attention_resolutions=[2, 4], channel_mult=[1, 2, 4], dims=3,
                   num_classes=None, num_heads=4)


def surface_points(n, seed, kind='sphere'):
    """Synthetic oriented point cloud in [-1, 1]^3: (points [n,3], unit normals [n,3]).  'sphere': radius 0.55 with
    a seed-dependent offset; 'torus': major 0.5 / minor 0.18.  A few points are snapped to cell boundaries of the
    depth-6 grid, duplicated, and put on the faces of the cube (the edge cases of the key computation)."""
    import math
    g = torch.Generator().manual_seed(seed)
    if kind == 'sphere':
        v = torch.randn(n, 3, generator=g)
        nrm = v / v.norm(dim=1, keepdim=True)
        c = (torch.rand(3, generator=g) - 0.5) * 0.3
        pts = nrm * 0.55 + c
    else:
        u = torch.rand(n, generator=g) * 2 * math.pi
        w = torch.rand(n, generator=g) * 2 * math.pi
        R, r = 0.5, 0.18
        pts = torch.stack([(R + r * torch.cos(w)) * torch.cos(u), (R + r * torch.cos(w)) * torch.sin(u), r * torch.sin(w)], 1)
        nrm = torch.stack([torch.cos(w) * torch.cos(u), torch.cos(w) * torch.sin(u), torch.sin(w)], 1)
    k = max(n // 50, 4)
    pts[:k] = torch.round(pts[:k] * 32) / 32              # exactly on depth-6 cell boundaries
    pts[k:2 * k] = pts[:k]                                # duplicates
    pts[2 * k] = torch.tensor([1.0, -1.0, 0.25])          # cube faces / corner
    pts[2 * k + 1] = torch.tensor([-1.0, -1.0, -1.0])
    pts[2 * k + 2] = torch.tensor([1.0, 1.0, 1.0])
    return pts.clamp(-1.0, 1.0).contiguous(), nrm.contiguous()
