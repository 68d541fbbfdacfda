"""The precision contract, with data.

north_star's bar is 1e-3 relative fp32 against the reference path.  Two figures per whole denoising step at the real
widths (tests/test_gpu_fullwidth.errors): rel-to-max, and the ELEMENT-WISE error |a-b| / max(|b|, 1 % of max|b|) at
its 99.9th percentile.  The reference's own fp32 arithmetic is not exact either: tools/oracle_noise.py runs the oracle
in float32 and in float64 on the same inputs (profiles/r03/oracle_fp32_noise.json: element-wise p99.9 2.0e-4 on an hr
step, 5.9e-4 on an lr step), so errors are taken against the float64 run of the oracle -- the reference's op sequence
without its rounding noise -- and the float32 run is measured beside the product.

Default mode = fp16x3: operands as fp16 hi + lo pairs (22 significand bits), three fp16 MFMAs per product, fp32
accumulate.  Round 2's bf16 pairs (16 bits) left an hr step at 1.7e-3 and an lr step at 8.9e-3 on the element-wise
figure (profiles/r03/precision_attribution.json); the fp16 pairs cost the same MFMAs.  Asserted here:
  * hr / lr step, default mode: element-wise p99.9 <= 1e-3 AND rel-to-max <= 5e-5;
  * the default mode is within 3x of the reference's own fp32 noise / of the exact-fp32 mode on the element-wise figure;
  * 50 DDIM steps: default mode vs exact fp32 vs the CPU oracle on the same initial noise -- the relative L2 distance of
    the default mode to the oracle stays below 1e-4 and within 3x of the exact-fp32 mode's.
"""
import json
import os

import pytest
import torch

import common as C
import oracle_cache as OC
from test_gpu_fullwidth import dev, errors, report, shell6_gpu

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _nets():
    from octfusion_amd import configs, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    net = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    return net.to(dev()).eval()


def _contract(step, run, case, p999_bar=1e-3):
    """Default mode and exact fp32 against the float64 oracle run; the float32 oracle run's own distance to it (the
    reference's rounding noise, computed on the full tensors when the fixture was made) beside them."""
    from octfusion_amd import ops
    ref64, floor = case['ref64'], case['floor']
    assert ops.get_precision() == 'fp16x3'
    e = errors(run(), ref64)
    ops.set_precision('fp32')
    try:
        e32 = errors(run(), ref64)
    finally:
        ops.set_precision(ops.DEFAULT_PRECISION)
    report(dict(test='precision_contract', step=step, default=e, exact_fp32=e32, reference_fp32_noise=floor))
    assert e['elementwise_p999'] <= p999_bar, (step, e)
    assert e['rel_to_max'] <= 5e-5, (step, e)
    assert e['elementwise_p999'] <= 3 * max(floor['elementwise_p999'], e32['elementwise_p999']), (step, e, floor, e32)
    return e, e32, floor


def test_elementwise_contract_hr_and_lr_steps():
    from octfusion_amd import sampler
    B = 2
    oc, doc = shell6_gpu(B)
    net = _nets()
    x = C.rand_input('fw_snet_uncond', doc.total_num, 3)
    log_snr = sampler.beta_linear_log_snr(torch.full((B,), 0.6))
    _contract('hr', lambda: net(unet_type='hr', x=x.to(dev()), doctree=doc, unet_lr=net.unet_lr,
                                timesteps=log_snr.to(dev()), x_self_cond=None, label=None), OC.get('hr_step_snet_uncond'))
    Bl = 4
    xl = C.rand_input('fw_lr', Bl, 8, 16, 16, 16)
    xsc = C.rand_input('fw_lr_sc', Bl, 8, 16, 16, 16)
    ls = sampler.beta_linear_log_snr(torch.full((Bl,), 0.3))
    _contract('lr', lambda: net(unet_type='lr', x=xl.to(dev()), timesteps=ls.to(dev()), x_self_cond=xsc.to(dev())),
              OC.get('lr_step'))


def test_elementwise_contract_feature_step():
    """the feature stage (obja_uncond, shell-8 B = 1, N = 448 232; graph_unet_hr.py:214-281 with the hr
    net as its middle, configs/octfusion_obja_uncond.yaml:11-24) on the same contract as hr and lr: element-wise p99.9 <=
    1e-3 and rel-to-max <= 5e-5 against the oracle in FLOAT64, with the float32 oracle's own distance to it reported.
    (Round 4 measured 2.1e-3 in every mode against the float32 oracle: that was the float32 reference's noise --
    scatter_add GroupNorm sums over 448 k rows in fp32 -- not the product's; `reference_fp32_noise` records it.)"""
    from octfusion_amd import configs, sampler, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    from test_gpu_fullwidth import shell8_gpu
    doc = shell8_gpu()
    net = UNet3DModel(**configs.unet_params('obja_uncond', 'feature'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    x = C.rand_input('fw_feature', doc.total_num, 3)
    log_snr = sampler.beta_linear_log_snr(torch.full((1,), 0.45))
    _contract('feature', lambda: net(unet_type='feature', x=x.to(dev()), doctree=doc, unet_lr=net.unet_hr,
                                     timesteps=log_snr.to(dev()), x_self_cond=None, label=None), OC.get('feature_step'))


def _drift_case():
    from octfusion_amd import configs
    from oracle import modules as OM, sampler as OS, unet as OU
    from test_gpu_fullwidth import _net_sd, shell6_oracle
    steps, B = 50, 1
    _, o_doc = shell6_oracle(1, False)
    sd = _net_sd('snet_uncond', 'hr')
    st = configs.stage_cfgs('snet_uncond')
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr')}
    init = C.rand_input('drift_init', o_doc.total_num, 3)
    x_or = OS.sample_loop(lambda x, ls, xs: OU.hr_forward(parts['unet_hr'], st['hr'], x, o_doc, ls, None, parts['unet_lr'],
                                                           st['lr']),
                          tuple(init.shape), B, steps, 'hr', 'eps', init_noise=init)
    return {'x_or': x_or, 'norm': float(x_or.double().norm())}


OC.register('ddim_drift_50', _drift_case)


def test_ddim_drift_50_steps():
    """50 eps-branch DDIM steps of the hr net (+ nested lr) on one shell-6 shape at the real widths, the same initial
    noise everywhere: default mode vs exact fp32 (both on the GPU) vs the CPU oracle (fixture `ddim_drift_50`)."""
    from octfusion_amd import ops, sampler
    steps, B = 50, 1
    oc, doc = shell6_gpu(B, jitter=False)
    net = _nets()
    init = C.rand_input('drift_init', doc.total_num, 3)

    def gpu_run():
        return sampler.sample_loop(net, tuple(init.shape), B, steps, 'hr', 'eps', dev(), doctree=doc, unet_lr=net.unet_lr,
                                   init_noise=init).cpu()
    x_def = gpu_run()
    ops.set_precision('fp32')
    try:
        x_f32 = gpu_run()
    finally:
        ops.set_precision(ops.DEFAULT_PRECISION)
    case = OC.get('ddim_drift_50')
    x_or, nrm = case['x_or'], case['norm']
    if isinstance(x_or, OC.Sketch):                   # (27 k x 3 values: above the verbatim threshold -> row sample)
        idx = x_or.idx.long()
        x_def_s, x_f32_s, x_or_s = x_def[idx], x_f32[idx], x_or.rows
        nrm_s = float(x_or_s.double().norm())
        d_impl = float((x_f32_s - x_or_s).norm()) / nrm_s
        d_def = float((x_def_s - x_or_s).norm()) / nrm_s
    else:
        d_impl = float((x_f32 - x_or).norm()) / nrm
        d_def = float((x_def - x_or).norm()) / nrm
    d_prec = float((x_def - x_f32).norm()) / nrm
    report(dict(test='ddim_drift', steps=steps, N=doc.total_num, default_vs_fp32=d_prec, fp32_vs_oracle=d_impl,
                default_vs_oracle=d_def))
    assert all(torch.isfinite(t).all() for t in (x_def, x_f32))
    # (measured in round 3 with bf16 pairs: 8.2e-6 vs 1.4e-6 for exact fp32 -- both three orders below the 1e-3 bar)
    assert d_def <= 1e-4, (d_def, d_impl)
    assert d_def <= 3.0 * d_impl + 2e-6, (d_def, d_impl)
