"""Oracle: DDIM loop + split<->octree conversions (torch CPU).  TEST INFRASTRUCTURE.

Restates reference models/octfusion_model_union.py:293-352 (sample_loop),
models/networks/diffusion_networks/ldm_diffusion_util.py:293-325 and
utils/util_dualoctree.py:225-273.
"""
import copy

import torch

from .octree import Octree

TRUNCATED_TIME = 0.7          # octfusion_model_union.py:39


def beta_linear_log_snr(t):
    """ldm_diffusion_util.py:301-302."""
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def log_snr_to_alpha_sigma(log_snr):
    """ldm_diffusion_util.py:308-309."""
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def get_sampling_timesteps(batch, steps):
    """octfusion_model_union.py:293-298: list of (t, t_next) pairs, each [batch]."""
    times = torch.linspace(1., 0., steps + 1)
    times = times.unsqueeze(0).repeat(batch, 1)
    return [(times[:, i], times[:, i + 1]) for i in range(steps)]


def _pad(x, t):
    return t.view(*t.shape, *((1,) * (x.ndim - t.ndim))) if x.ndim > t.ndim else t


def sample_loop(net, shape, batch_size, ddim_steps, unet_type, df_type, truncated_index=0.0,
                init_noise=None, step_noise=None):
    """octfusion_model_union.py:300-352.

    ``net(x, noise_cond, x_self_cond)`` is the denoiser call; noise is drawn from the
    global torch RNG exactly where the reference draws it unless explicit tensors are
    supplied (``init_noise`` [shape], ``step_noise`` list of [shape], x0 branch only).
    """
    x = torch.randn(shape) if init_noise is None else init_noise.clone()
    x_start = None
    for i, (t, t_next) in enumerate(get_sampling_timesteps(batch_size, ddim_steps)):
        log_snr = beta_linear_log_snr(t)
        log_snr_next = beta_linear_log_snr(t_next)
        output = net(x, log_snr, x_start)
        if t[0] < truncated_index and unet_type == 'lr':
            output.sign_()
        if df_type == 'x0':
            x_start = output
            pl, pln = _pad(x, log_snr), _pad(x, log_snr_next)
            alpha, sigma = log_snr_to_alpha_sigma(pl)
            alpha_next, sigma_next = log_snr_to_alpha_sigma(pln)
            c = -torch.special.expm1(pl - pln)
            mean = alpha_next * (x * (1 - c) / alpha + c * output)
            variance = (sigma_next ** 2) * c
            rnd = torch.randn_like(x) if step_noise is None else step_noise[i]
            noise = torch.where(_pad(x, t_next > truncated_index), rnd, torch.zeros_like(x))
            x = mean + torch.sqrt(variance) * noise
        elif df_type == 'eps':
            alpha, sigma = log_snr_to_alpha_sigma(log_snr)
            alpha_next, sigma_next = log_snr_to_alpha_sigma(log_snr_next)
            alpha, sigma, alpha_next, sigma_next = alpha[0], sigma[0], alpha_next[0], sigma_next[0]
            x_start = (x - output * sigma) / alpha.clamp(min=1e-8)
            x = x_start * alpha_next + output * sigma_next
    return x


def create_full_octree(depth, full_depth, batch_size):
    """ldm_diffusion_util.py:318-325."""
    octree = Octree(depth, full_depth, batch_size)
    for d in range(full_depth + 1):
        octree.octree_grow_full(d)
    octree.depth = full_depth
    return octree


def split2octree_small(split, input_depth, full_depth):
    """util_dualoctree.py:225-250."""
    ds = (split > 0).to(split.dtype)
    B = ds.shape[0]
    octree = create_full_octree(input_depth, full_depth, B)
    nempty_vox = ds.sum(dim=1) > 0
    x, y, z, b = octree.xyzb(full_depth)
    octree.octree_split(nempty_vox[b, x, y, z].long(), full_depth)
    octree.octree_grow(full_depth + 1)
    octree.depth += 1
    x, y, z, b = octree.xyzb(full_depth, nempty=True)
    label = ds[b, :, x, y, z].reshape(-1).long()
    octree.octree_split(label, full_depth + 1)
    octree.octree_grow(full_depth + 2)
    octree.depth += 1
    return octree


def split2octree_large(octree, split, small_depth):
    """util_dualoctree.py:252-273."""
    ds = (split > 0).to(split.dtype)
    out = copy.deepcopy(octree)
    ssum = ds.sum(dim=1)
    out.octree_split((ssum > 0).long(), small_depth)
    out.octree_grow(small_depth + 1)
    out.depth += 1
    label = ds[ssum > 0].reshape(-1).long()
    out.octree_split(label, small_depth + 1)
    out.octree_grow(small_depth + 2)
    out.depth += 1
    return out


def octree2split_small(octree, full_depth):
    """util_dualoctree.py:199-211: [B, 8, S, S, S] in {-1, +1}; channel j of cell (x, y, z) = child j of that
    depth-full_depth node is non-empty."""
    from .octree import octree2voxel, octree_pad
    nz = (octree.children[full_depth + 1] >= 0).reshape(-1, 8)
    pad = octree_pad(nz, octree, full_depth)
    vox = octree2voxel(pad, octree, full_depth).permute(0, 4, 1, 2, 3).contiguous()
    return 2 * vox.float() - 1


def octree2split_large(octree, small_depth):
    """util_dualoctree.py:213-223: [nnum[small_depth], 8] in {-1, +1}."""
    from .octree import octree_pad
    nz = (octree.children[small_depth + 1] >= 0).reshape(-1, 8)
    return 2 * octree_pad(nz, octree, small_depth).float() - 1
