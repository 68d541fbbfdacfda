"""DDIM sampling driver: the loop that calls the U-Net once per denoising step.

Own code reproducing the semantics of reference
models/octfusion_model_union.py:293-352 (``sample_loop``): continuous-time
log-SNR schedule (ldm_diffusion_util.py:293-309), x0 branch with the 0.7
truncation + sign_() for the "lr" stage, eps branch for "hr" / "feature".  The
elementwise updates are libofx kernels whose coefficients live in device memory
(so a step can be replayed from a hipGraph); the per-step scalars are computed
on the host in fp32 exactly as the reference computes them.
"""
import torch

from . import ops

TRUNCATED_TIME = 0.7      # octfusion_model_union.py:39


def beta_linear_log_snr(t):
    return -torch.log(torch.special.expm1(1e-4 + 10 * (t ** 2)))


def log_snr_to_alpha_sigma(log_snr):
    return torch.sqrt(torch.sigmoid(log_snr)), torch.sqrt(torch.sigmoid(-log_snr))


def sampling_times(steps):
    """[(t, t_next)] for linspace(1, 0, steps+1) (octfusion_model_union.py:293-298)."""
    times = torch.linspace(1., 0., steps + 1)
    return [(times[i], times[i + 1]) for i in range(steps)]


def eps_coef(t, t_next):
    a, s = log_snr_to_alpha_sigma(beta_linear_log_snr(t))
    an, sn = log_snr_to_alpha_sigma(beta_linear_log_snr(t_next))
    return torch.stack([a, s, an, sn]).float()


def x0_coef(t, t_next, truncated_index):
    ls, lsn = beta_linear_log_snr(t), beta_linear_log_snr(t_next)
    a, s = log_snr_to_alpha_sigma(ls)
    an, sn = log_snr_to_alpha_sigma(lsn)
    c = -torch.special.expm1(ls - lsn)
    sd = torch.sqrt((sn ** 2) * c) if bool(t_next > truncated_index) else torch.zeros(())
    return torch.stack([a, c, an, sd]).float()


def ddim_eps_step(x, eps, t, t_next):
    coef = eps_coef(torch.as_tensor(t, dtype=torch.float32), torch.as_tensor(t_next, dtype=torch.float32))
    return ops.ddim_eps_update(x, eps.contiguous(), coef.to(x.device))


@torch.no_grad()
def sample_loop(net, shape, batch_size, ddim_steps, unet_type, df_type, device, doctree=None, unet_lr=None,
                label=None, truncated_index=0.0, init_noise=None, step_noise=None):
    """Run `ddim_steps` denoising steps; `net` is a graph_unet_union.UNet3DModel (or any callable with
    its keyword interface).  Noise comes from torch's device RNG unless given explicitly."""
    x = torch.randn(shape, device=device) if init_noise is None else init_noise.to(device).clone()
    x = x.contiguous()
    x_start = None
    for i, (t, t_next) in enumerate(sampling_times(ddim_steps)):
        noise_cond = beta_linear_log_snr(t).float().expand(batch_size).contiguous().to(device)
        out = net(unet_type=unet_type, x=x, doctree=doctree, timesteps=noise_cond, unet_lr=unet_lr,
                  x_self_cond=x_start, label=label)
        if float(t) < truncated_index and unet_type == 'lr':
            out = out.sign_()
        out = out.contiguous()
        if df_type == 'x0':
            x_start = out
            coef = x0_coef(t, t_next, truncated_index).to(device)
            noise = None
            if float(coef[3]) != 0.0:
                noise = torch.randn_like(x) if step_noise is None else step_noise[i].to(device)
            ops.ddim_x0_update(x, out, noise, coef)
        elif df_type == 'eps':
            # the reference also keeps x_start = (x - eps*sigma)/alpha and passes it on as x_self_cond
            # (octfusion_model_union.py:349, :320); the hr / feature nets ignore it, so it is only
            # materialised for nets that declare they want it.
            x0_out = torch.empty_like(x) if getattr(net, 'wants_self_cond', True) else None
            ops.ddim_eps_update(x, out, eps_coef(t, t_next).to(device), x0_out)
            x_start = x0_out
        else:
            raise ValueError(df_type)
    return x
