"""Oracle: VAE training objective (torch CPU, autograd).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Restates reference models/networks/dualoctree_networks/loss.py:12-29 (compute_gradient, sdf_reg_loss), :100-122
(compute_mpu_gradients, compute_octree_loss), :124-133 (compute_sdf_loss), :164-178 (geometry_loss) and the
posterior of distributions.py:24-46.  The MPU gradient is autograd through oracle/mpu.py with the reference's
|.| convention (mpu.py:18-32: derivative +1 at 0), create_graph=True so the gradient loss reaches the codes.
Pinned against the reference's own outputs through tests/golden/g_vae_train.pt.
"""
import torch
import torch.nn.functional as F

from . import mpu as OM


def compute_gradient(y, x):
    """loss.py:12-20."""
    return torch.autograd.grad(y, [x], torch.ones_like(y), create_graph=True)[0]


def mpu_gradients(mpus, pos):
    """loss.py:100-108: {d: d fval / d pos[:, :3]}."""
    return {d: compute_gradient(fval, pos)[:, :3] for d, (fval, _) in mpus.items()}


def sdf_reg_loss(sdf, grad, sdf_gt, grad_gt, suffix=''):
    """loss.py:23-29."""
    return {'grad_loss' + suffix: (grad - grad_gt).pow(2).mean() * 1.0,
            'sdf_loss' + suffix: (sdf - sdf_gt).pow(2).mean() * 200.0}


def octree_loss(logits, octree):
    """loss.py:110-122."""
    out = {}
    for d, logit in logits.items():
        label = octree.nempty_mask(d).long()
        out['loss_%d' % d] = F.cross_entropy(logit, label)
        out['accu_%d' % d] = logit.argmax(1).eq(label).float().mean()
    return out


def posterior(params, noise):
    """distributions.py:24-46: (z, kl elementwise)."""
    mean, logvar = torch.chunk(params, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    z = mean + torch.exp(0.5 * logvar) * noise
    return z, 0.5 * (mean.pow(2) + torch.exp(logvar) - 1.0 - logvar)


def geometry_loss(logits, mpus, octree, pos, sdf_gt, grad_gt, kl=None, kl_weight=1.0):
    """loss.py:164-178 with reg_loss_type 'sdf_reg_loss'; `mpus` must have been evaluated at `pos`
    (requires_grad).  Returns the dict of named losses / accuracies."""
    out = octree_loss(logits, octree)
    grads = mpu_gradients(mpus, pos)
    for d, (sdf, _) in mpus.items():
        out.update(sdf_reg_loss(sdf, grads[d], sdf_gt, grad_gt, '_%d' % d))
    if kl is not None:
        out['kl_loss'] = kl_weight * kl.mean()
    return out


def total_loss(losses):
    """octfusion_model_vae.py:182-183: the sum of every entry whose name contains 'loss'."""
    return torch.sum(torch.stack([v for k, v in losses.items() if 'loss' in k]))
