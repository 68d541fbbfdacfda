"""Deterministic synthetic workloads (no dataset / checkpoint exists offline).

shell-6 / shell-8 sphere-shell octrees of SURVEY.md section 8(d), plus random-but-seeded
weights: every parameter ~ N(0, 1.5/sqrt(fan_in)) (zero-initialised modules included, so
every branch carries signal).  Pure torch; used by bench.py, smoke() and the tests.
"""
import zlib

import torch


def shell6_split(B, jitter=True):
    """[B, 8, 16, 16, 16] split codes: depth-5 occupancy 9 < |p - 15.5| < 11 on 32^3
    (+0.25*(b mod 4) radius jitter so batch elements are not identical)."""
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
    """[nnum6, 8] split codes: depth-7 occupancy 38.5 < |p - 63.5| < 41 on 128^3."""
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


def fill_param(key, shape):
    g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
    shape = tuple(shape)
    t = torch.randn(shape, generator=g)
    eff = [s for s in shape if s != 1]
    if len(eff) <= 1:
        if key.endswith('weight') or key.endswith('weights'):
            return t if 'time_pos_emb' in key else 1.0 + 0.1 * t
        return 0.1 * t
    fan = t.numel() // (shape[1] if key.endswith('.weights') and len(shape) == 2 else shape[0])
    return t * (1.5 / max(1, fan) ** 0.5)


def random_state_dict(module):
    return {k: fill_param(k, v.shape) for k, v in module.state_dict().items()}
