"""fp16x3's range, guarded and tested (include/ofx.h "fp16x3 range guard").

The default contraction carries operands as fp16 hi + lo pairs.  Reference arithmetic to match: fp32
(modules.py:194-220).  What is planted here, against the oracle in float64 with the element-wise metric:
  * weight tensors far below O(1) -- a whole tensor at 1e-4 scale (a zero-initialised projection that learned small
    weights) and weights log-uniform in [1e-6, 1]: per-tensor power-of-two scaling at pack time keeps them fp32-class
    (without it fp16's denormal floor of 3e-8 is a 3e-4 RELATIVE error on a 1e-4-scale tensor);
  * activations up to 6e4 (inside the range) on the planes GraphConv and on a 1x1 convolution of an un-normalised
    tensor;
  * activations up to 1e5: operands beyond +-65504 are not clamped -- they poison the result with NaN (planes
    GraphConv, where ofx_planes_split also counts them exactly, and the register-staged GEMM alike);
    ops.raise_on_range_error raises and switches the process to bf16x3, in which the same layers then pass at
    bf16x3's accuracy;
  * a whole hr step with the synthetic weights does NOT trip it.
"""
import math

import pytest
import torch

import common as C
from test_gpu_fullwidth import dev, errors, report, shell6

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _graph(B=2):
    oc, doc, o_oc, o_doc = shell6(B)
    return doc, o_doc


def _conv_pair(cin, cout, d, weights):
    from octfusion_amd import modules as M
    conv = M.GraphConv(cin, cout, 7, 7, d - 1)
    conv.load_state_dict({'weights': weights})
    return conv.to(dev())


def _run_planes(conv, x, doc, d):
    from octfusion_amd import ops
    saved = ops.PLANES_MIN_TILES
    ops.PLANES_MIN_TILES = 1
    try:
        return conv(x.to(dev()), doc, d, split_input=True)
    finally:
        ops.PLANES_MIN_TILES = saved


def _clean(ops):
    ops.range_words(dev()).zero_()
    ops.set_precision(ops.DEFAULT_PRECISION)


def test_small_weight_tensors_stay_fp32_class():
    from octfusion_amd import ops
    from oracle import modules as OM
    doc, o_doc = _graph()
    d, cin, cout = 5, 128, 128
    N = doc.csr(d)[2]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, cin, generator=g)
    K = 7 * (cin + d - 1)
    cases = {
        'tensor_at_1e-4': 1e-4 * torch.randn(K, cout, generator=g),
        'log_uniform_1e-6_to_1': torch.exp(torch.empty(K, cout).uniform_(math.log(1e-6), 0.0, generator=g)) *
                                 (torch.randint(0, 2, (K, cout), generator=g) * 2 - 1).float(),
        'tensor_at_1e3': 1e3 * torch.randn(K, cout, generator=g),
    }
    _clean(ops)
    for name, w in cases.items():
        conv = _conv_pair(cin, cout, d, w)
        ref = OM.graph_conv(x.double(), o_doc, d, w.double(), None, d - 1)
        ref32 = OM.graph_conv(x, o_doc, d, w, None, d - 1)
        floor = errors(ref32, ref)                              # the reference's own fp32 arithmetic
        e = errors(_run_planes(conv, x, doc, d), ref)           # planes kernel, fp16 pairs
        ops.USE_PLANES = False
        try:
            e_reg = errors(conv(x.to(dev()), doc, d), ref)      # register-staged kernel, fp16 pairs
        finally:
            ops.USE_PLANES = True
        report(dict(test='range_small_weights', case=name, planes=e, register_staged=e_reg, reference_fp32_noise=floor))
        for ee in (e, e_reg):
            assert ee['elementwise_p999'] <= max(3 * floor['elementwise_p999'], 2e-5), (name, ee, floor)
            assert ee['rel_to_max'] <= 2e-5, (name, ee)
    assert ops.range_error(dev()) is None


def test_dense_gemm_small_weights_and_unnormalised_activations():
    """Conv1x1 / Downsample-style GEMM (register-staged kernel, operands split in the kernel): weights at 1e-4 scale,
    activations with |x| up to 6e4 -- inside the range, so exact to fp32 class and no trip."""
    from octfusion_amd import modules as M, ops
    g = torch.Generator().manual_seed(5)
    n, cin, cout = 4096, 384, 256
    x = torch.randn(n, cin, generator=g) * 30.0
    x[::97, ::5] = 6.0e4 * torch.sign(x[::97, ::5])
    w = 1e-4 * torch.randn(cout, cin, generator=g)
    lin = M.Conv1x1(cin, cout)
    lin.load_state_dict({'linear.weight': w})
    lin = lin.to(dev())
    _clean(ops)
    ref = x.double() @ w.double().t()
    floor = errors(x @ w.t(), ref)
    e = errors(lin(x.to(dev())), ref)
    report(dict(test='range_dense_gemm', default=e, reference_fp32_noise=floor))
    assert e['elementwise_p999'] <= max(3 * floor['elementwise_p999'], 2e-5), (e, floor)
    assert ops.range_error(dev()) is None


def test_guard_trips_beyond_the_fp16_range_and_falls_back():
    from octfusion_amd import _lib, modules as M, ops
    from oracle import modules as OM
    doc, o_doc = _graph()
    d, cin, cout = 5, 128, 128
    N = doc.csr(d)[2]
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, cin, generator=g) * 100.0
    x[5::211, 3::17] = 1.0e5                                       # beyond +-65504
    w = C.fill_state_dict([('weights', (7 * (cin + d - 1), cout))])['weights']
    conv = _conv_pair(cin, cout, d, w)
    ref = OM.graph_conv(x.double(), o_doc, d, w.double(), None, d - 1)
    _clean(ops)
    try:
        y16 = _run_planes(conv, x, doc, d)                          # ofx_planes_split counts, the result is poisoned
        err = ops.range_error(dev(), y16)
        assert err == {'operands_beyond_fp16': int((x.abs() > 65504).sum()), 'result_non_finite': True}, err
        with pytest.raises(ops.OfxRangeError):
            ops.raise_on_range_error(dev(), y16)
        assert ops.get_precision() == 'bf16x3' and ops.range_error(dev()) is None
        e = errors(_run_planes(conv, x, doc, d), ref)               # the retry: bf16 pairs have fp32's range
        report(dict(test='range_fallback_bf16x3', **e))
        assert e['rel_to_max'] <= 2e-4, e
        assert ops.range_error(dev()) is None
        # the register-staged GEMM (operands split inside the kernel: 1x1 skip convolutions, pool / unpool): same rule
        ops.set_precision('fp16x3')
        lin = M.Conv1x1(cin, cout)
        wl = C.fill_state_dict([('linear.weight', (cout, cin))])['linear.weight']
        lin.load_state_dict({'linear.weight': wl})
        lin = lin.to(dev())
        yl = lin(x.to(dev()))
        assert not bool(torch.isfinite(yl).all())
        with pytest.raises(ops.OfxRangeError):
            ops.raise_on_range_error(dev(), yl)
        el = errors(lin(x.to(dev())), x.double() @ wl.double().t())
        assert el['rel_to_max'] <= 2e-4, el
        # in any other precision a non-finite result is the model's own business, not a range error
        with pytest.raises(_lib.OfxError) as ei:
            ops.raise_on_range_error(dev(), yl)
        assert not isinstance(ei.value, ops.OfxRangeError)
    finally:
        _clean(ops)


def test_a_whole_step_at_the_real_widths_does_not_trip():
    from octfusion_amd import configs, ops, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    from oracle import sampler as OS
    B = 2
    oc, doc, o_oc, o_doc = shell6(B)
    net = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    _clean(ops)
    x = C.rand_input('fw_snet_uncond', doc.total_num, 3)
    y = net(unet_type='hr', x=x.to(dev()), doctree=doc, unet_lr=net.unet_lr,
            timesteps=OS.beta_linear_log_snr(torch.full((B,), 0.6)).to(dev()), x_self_cond=None, label=None)
    assert bool(torch.isfinite(y).all())
    assert ops.range_error(dev()) is None
    ops.raise_on_range_error(dev())                                 # no-op
