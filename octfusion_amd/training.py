"""Training step of the first (lr) diffusion stage -- SURVEY 8f-4.

Mirror of the reference's optimize_parameters for stage_flag == "lr" (octfusion_model_union.py:242-269, 271-292,
478-487): noise the split codes at a random log-SNR, predict x0, MSE, backward, AdamW, EMA.  Forward and backward
run on libofx (octfusion_amd/backward.py); the optimiser and EMA are one elementwise kernel launch per parameter.
"""
import torch

from . import backward as BW
from . import ops, sampler
from ._lib import call, ptr, stream


class AdamW:
    """torch.optim.AdamW (the reference's optimiser, octfusion_model_union.py:142) on ofx_adamw_step."""

    def __init__(self, named_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, sync=True):
        self.params = dict(named_params)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {k: (torch.zeros_like(p.data), torch.zeros_like(p.data)) for k, p in self.params.items()}
        if sync:
            self.sync_()

    @torch.no_grad()
    def sync_(self, src=0):
        """Under torch.distributed: every rank takes rank `src`'s parameters and optimiser state (what wrapping the net
        in DistributedDataParallel does at construction for the reference, octfusion_model_union.py:185-196) -- ranks
        that were seeded differently would otherwise average gradients of different models, silently."""
        import torch.distributed as td
        from . import dist as D
        if not D._active():
            return 0
        keys = sorted(self.params)
        tensors = [self.params[k].data for k in keys] + [s for k in keys for s in self.state[k]]
        flat = torch.cat([t_.reshape(-1).float() for t_ in tensors])
        td.broadcast(flat, src=src)
        off = 0
        for t_ in tensors:
            n = t_.numel()
            t_.copy_(flat[off:off + n].view_as(t_))
            off += n
        for k in keys:
            _bump(self.params[k])
        step = torch.tensor([self.step_count], dtype=torch.int64, device=flat.device)
        td.broadcast(step, src=src)
        self.step_count = int(step.item())
        return flat.numel() * 4

    @torch.no_grad()
    def step(self, grads):
        """One AdamW update.  Under torch.distributed (one process per GPU) the gradients are first averaged over the
        ranks (dist.all_reduce_mean_: what DistributedDataParallel does for the reference)."""
        from . import dist as D
        D.all_reduce_mean_(grads)
        self.step_count += 1
        for k, p in self.params.items():
            g = grads.get(k)
            if g is None:         # no gradient this step: torch.optim skips the parameter entirely (no decay either)
                continue
            g = g.contiguous()
            assert g.shape == p.shape, (k, g.shape, p.shape)
            m, v = self.state[k]
            call('ofx_adamw_step', ptr(p.data), ptr(g), ptr(m), ptr(v), p.numel(), self.lr, self.betas[0], self.betas[1],
                 self.eps, self.weight_decay, self.step_count, stream())
            _bump(p)              # the kernel wrote behind torch's back: bump the version so packed-weight caches repack

    def state_dict(self):
        """{'step', 'state': {name: (exp_avg, exp_avg_sq)}} -- what checkpoint.save_ckpt stores under 'opt'."""
        return {'step': self.step_count, 'state': {k: (m.clone(), v.clone()) for k, (m, v) in self.state.items()}}

    def load_state_dict(self, sd):
        self.step_count = int(sd.get('step', 0))
        for k, (m, v) in sd.get('state', {}).items():
            if k in self.state:
                self.state[k][0].copy_(m)
                self.state[k][1].copy_(v)


def _bump(p):
    """Mark a tensor written by a libofx kernel as modified (what an in-place torch op would do) without
    launching one."""
    torch._C._increment_version(p)


def trainable_parameters(net, stage):
    """The parameters the reference optimises in `stage` ('lr' | 'hr' | 'feature'): only that stage's own net --
    the earlier stages are frozen (requires_grad False, octfusion_model_union.py:127-142,
    octfusion_model_union_3t.py:53-72) and AdamW is built from the trainable ones only."""
    prefix = {'lr': 'unet_lr.', 'hr': 'unet_hr.', 'feature': 'unet_feature.'}[stage]
    return {k: p for k, p in net.named_parameters() if k.startswith(prefix)}


@torch.no_grad()
def ema_update(ema_module, module, beta):
    """ldm_diffusion_util.py:50-53."""
    for pe, p in zip(ema_module.parameters(), module.parameters()):
        call('ofx_ema_update', ptr(pe.data), ptr(p.data), p.numel(), beta, stream())
        _bump(pe)


@torch.no_grad()
def lr_stage_step(net, opt, split_small, times=None, noise=None, label=None, ema=None, ema_rate=0.999):
    """One optimisation step of the lr stage on a batch of split codes [B, 8, S, S, S] (values in {-1, +1}).
    net: graph_unet_lr.UNet3DModel.  Returns the loss (python float)."""
    B = split_small.shape[0]
    dev = split_small.device
    if times is None:
        times = torch.rand(B, device=dev)
    if noise is None:
        noise = torch.randn_like(split_small)
    log_snr = sampler.beta_linear_log_snr(times.cpu()).float().to(dev)
    alpha, sigma = sampler.log_snr_to_alpha_sigma(log_snr)
    noised = alpha.view(B, 1, 1, 1, 1) * split_small + sigma.view(B, 1, 1, 1, 1) * noise
    # the lr net takes cat(x, x_self_cond); training passes no self-conditioning (zeros), graph_unet_lr.py:198-204
    x = torch.cat((noised, torch.zeros_like(noised)), dim=1)
    rows = ops.voxel2octree_cf(x.float(), net.full_depth)
    target = ops.voxel2octree_cf(split_small.float(), net.full_depth)
    box = {}

    def dy_fn(y):
        diff = y - target
        box['loss'] = float((diff * diff).mean())
        return diff * (2.0 / diff.numel())
    _, _, grads = BW.lr_unet_forward_backward(net, rows, B, log_snr, dy_fn, label=label)
    opt.step(grads)
    if ema is not None:
        ema_update(ema, net, ema_rate)
    return box['loss']


@torch.no_grad()
def hr_stage_step(net, opt, codes, doctree, depth, times=None, noise=None, label=None, ema=None, ema_rate=0.999,
                  stage='hr', df_type='eps'):
    """One optimisation step of a sparse stage (octfusion_model_union.py:242-269 / octfusion_model_union_3t.py):
    stage 'hr' = the hr net with the lr net nested, stage 'feature' = the feature net with the hr net nested;
    df_type 'eps' regresses the noise, 'x0' the clean data.  data [N, C] lives on the depth-`depth` dual graph
    (latent codes from GraphVAE.encode, or split codes); net = the union UNet3DModel; `opt` holds the parameters
    under their union state_dict names.  Returns the loss."""
    B = doctree.batch_size
    dev = codes.device
    if times is None:
        times = torch.rand(B, device=dev)
    if noise is None:
        noise = torch.randn_like(codes)
    log_snr = sampler.beta_linear_log_snr(times.cpu()).float().to(dev)
    alpha, sigma = sampler.log_snr_to_alpha_sigma(log_snr)
    bid = doctree.batch_id(depth)
    noised = (alpha[bid].unsqueeze(1) * codes + sigma[bid].unsqueeze(1) * noise).contiguous()
    box = {}

    target = noise if df_type == 'eps' else codes

    def dy_fn(y):
        diff = y - target
        box['loss'] = float((diff * diff).mean())
        return diff * (2.0 / diff.numel())
    outer, nested = ('unet_hr', 'unet_lr') if stage == 'hr' else ('unet_feature', 'unet_hr')
    _, _, g_o, _g_nested = BW.hr_unet_forward_backward(getattr(net, outer), noised, doctree, getattr(net, nested),
                                                      log_snr, dy_fn, label=label)
    # only the stage's own net is trained: the nested earlier-stage net is frozen in the reference
    # (octfusion_model_union.py:127-142, octfusion_model_union_3t.py:53-72), so its gradients are dropped and
    # parameters without a gradient are skipped by the optimiser (torch's grad=None behaviour: no update, no decay)
    grads = {outer + '.' + k: v for k, v in g_o.items()}
    opt.step(grads)
    if ema is not None:
        ema_update(ema, net, ema_rate)
    return box['loss']
