"""GPU parity of the cascade orchestration and of the 3-stage checkpoint layout (8f-3).

The three stages feed each other through DISCRETE octrees (sign of the lr split codes, split > 0 of the hr output):
a 1e-6 difference at a value near zero would grow a different tree and make the later stages incomparable.  So the
test runs the product's CascadeSampler with explicit noises once, and replays every stage in the oracle
(oracle.sampler.sample_loop driven by the oracle's functional nets over the same state_dict) on the octree the
product's previous stage produced -- each stage's output is compared with the oracle's, and the number of
tree-deciding sign disagreements is bounded.
Reference: models/octfusion_model_union_3t.py:152-214 (3 stages), models/octfusion_model_union.py:354-401 (2 stages),
seeding :372,:390.
"""
import os
import tempfile

import pytest
import torch

import common as C

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def dev():
    return torch.device('cuda:0')


CFG3 = dict(image_size=[8, 32, 128], input_depth=[3, 5, 7], unet_type=['lr', 'hr', 'feature'], full_depth=3,
            input_channels=[8, 8, 3], out_channels=[8, 8, 3], model_channels=[16, 32, 32],
            num_res_blocks=[[1, 1, 1], [1, 1, 0], [1, 1, 1]], attention_resolutions=[2, 4],
            channel_mult=[[1, 2, 4], [1, 2, 4], [1, 2, 4]], num_heads=4, use_checkpoint=False, dims=3,
            df_type=['x0', 'x0', 'x0'])


def stage_cfgs(p):
    out = {}
    for i, kind in enumerate(p['unet_type']):
        if kind == 'lr':
            out[kind] = dict(kind='lr', full_depth=p['full_depth'], model_channels=p['model_channels'][i],
                             channel_mult=p['channel_mult'][i], attention_resolutions=p['attention_resolutions'],
                             num_heads=p['num_heads'], num_classes=None)
        else:
            out[kind] = dict(kind='hr', input_depth=p['input_depth'][i], full_depth=p['full_depth'],
                             model_channels=p['model_channels'][i], channel_mult=p['channel_mult'][i],
                             num_res_blocks=p['num_res_blocks'][i], num_classes=None)
    return out


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-9))


class Recorder:
    """Wraps the union net: records every denoiser call (inputs and a copy of the output before the sampler's in-place
    sign_()) so that each step can be replayed in the oracle on the product's own inputs."""

    def __init__(self, net):
        self._net = net
        self.calls = []

    def __getattr__(self, name):
        return getattr(self._net, name)

    def parameters(self):
        return self._net.parameters()

    def __call__(self, unet_type=None, x=None, doctree=None, timesteps=None, unet_lr=None, x_self_cond=None, label=None):
        y = self._net(unet_type=unet_type, x=x, doctree=doctree, timesteps=timesteps, unet_lr=unet_lr,
                      x_self_cond=x_self_cond, label=label)
        self.calls.append(dict(stage=unet_type, x=x.clone().cpu(), t=timesteps.clone().cpu(),
                               xsc=None if x_self_cond is None else x_self_cond.clone().cpu(), y=y.clone().cpu(),
                               doctree=doctree))
        return y


def test_three_stage_cascade_vs_oracle():
    """All three stages through CascadeSampler with explicit noises; every denoiser call and every DDIM update is
    replayed in the oracle ON THE PRODUCT'S OWN STEP INPUT.  (A whole-loop comparison is meaningless here: with
    random weights the tiny nets are chaotic -- a 1e-5 perturbation of the initial noise flips 43 % of the signs of the
    ORACLE's own lr result after 4 steps -- so the steps are checked one by one.)"""
    worst = _three_stage_cascade(1e-3)
    print('cascade per-call rel-to-max errors vs oracle:', worst)


def test_cascade_in_fp16_single_pass():
    """BASELINE configs[4] ("depth-8/6/4 cascade ... fp16 MFMA"): the same three-stage cascade with the planes GraphConv
    in its single-pass fp16 mode (`ops.set_precision('fp16')`: one v_mfma_f32_32x32x16_f16 per product, activations and
    weights rounded to fp16; every other contraction in bf16 pairs), every denoiser call against the fp32 oracle.
    STATED BOUND: 2e-2 rel-to-max per call (measured here: ~1e-3 ... 6e-3; at the real widths the whole feature step
    reads 6.2e-4 rel-to-max and 2e-2 element-wise p99.9) -- reduced precision by construction, which
    is why the shipped default of every stage, cfg5 included, is fp16x3."""
    from octfusion_amd import ops
    saved = (ops.get_precision(), ops.PLANES_MIN_TILES)
    ops.set_precision('fp16')
    ops.PLANES_MIN_TILES = 1                       # the tiny fixture layers must take the planes kernel too
    try:
        worst = _three_stage_cascade(2e-2, steps=2)      # (two steps per stage: the float64 oracle replay is the test's time)
    finally:
        ops.set_precision(saved[0])
        ops.PLANES_MIN_TILES = saved[1]
    print('fp16 single-pass cascade, per-call rel-to-max errors vs oracle:', worst)
    from test_gpu_fullwidth import report
    report(dict(test='cascade_fp16_single_pass', bound=2e-2, **{'rel_to_max_' + k: v for k, v in worst.items()}))


def _three_stage_cascade(tol, steps=3):
    from octfusion_amd import pipeline, sampler
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.graph_unet_union import UNet3DModel
    from octfusion_amd.octree import split2octree_large, split2octree_small
    from oracle import dual_octree as OD, modules as OM, sampler as OS, unet as OU
    B = 1
    net = UNet3DModel(**{k: v for k, v in dict(CFG3, stage_flag='feature').items() if k != 'df_type'})
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()])
    net.load_state_dict(sd)
    net = net.to(dev()).eval()
    st = stage_cfgs(CFG3)
    parts = {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr', 'unet_feature')}

    g = torch.Generator().manual_seed(5)
    S = 1 << CFG3['full_depth']
    shape_lr = (B, 8, S, S, S)
    n_lr = dict(init=torch.randn(shape_lr, generator=g), steps=[torch.randn(shape_lr, generator=g) for _ in range(steps)])
    # node counts of the later stages are only known after the earlier ones: their noises are drawn then
    # the first two stages by hand (to learn the node counts the noises must be cut to) ...
    split_small = sampler.sample_loop(net, shape_lr, B, steps, 'lr', 'x0', dev(), truncated_index=sampler.TRUNCATED_TIME,
                                      init_noise=n_lr['init'], step_noise=n_lr['steps'], use_graph=False)
    d_small = CFG3['input_depth'][1]
    oc5 = split2octree_small(split_small, d_small, CFG3['full_depth'])
    n5 = DualOctree(oc5).total_num
    n_hr = dict(init=torch.randn(n5, 8, generator=g), steps=[torch.randn(n5, 8, generator=g) for _ in range(steps)])
    x_hr = sampler.sample_loop(net, (n5, 8), B, steps, 'hr', 'x0', dev(), doctree=DualOctree(oc5), unet_lr=net.unet_lr,
                               init_noise=n_hr['init'], step_noise=n_hr['steps'], use_graph=False)
    nn5 = int(oc5.nnum[d_small])
    oc7 = split2octree_large(oc5, x_hr[x_hr.shape[0] - nn5:].contiguous(), d_small)
    n7 = DualOctree(oc7).total_num
    n_ft = dict(init=torch.randn(n7, 3, generator=g), steps=[torch.randn(n7, 3, generator=g) for _ in range(steps)])

    # ... then the orchestration under test: CascadeSampler.sample runs all three stages with the same noises
    rec = Recorder(net)
    cs = pipeline.CascadeSampler(rec, CFG3, None)
    noises = {'lr': n_lr, 'hr': n_hr, 'feature': n_ft}
    res = cs.sample(B, ddim_steps=steps, noises=noises, use_graph=False)
    assert set(res) >= {'split_small', 'octree_small', 'hr', 'octree_large', 'feature', 'doctree'}
    assert torch.equal(res['split_small'], split_small)              # same kernels, same noises: bit-equal
    assert torch.equal(res['hr'], x_hr)
    assert res['octree_small'].nnum.tolist() == oc5.nnum.tolist()
    assert res['octree_large'].nnum.tolist() == oc7.nnum.tolist()
    assert res['feature'].shape == (n7, 3) and bool(torch.isfinite(res['feature']).all())
    assert [c['stage'] for c in rec.calls] == ['lr'] * steps + ['hr'] * steps + ['feature'] * steps

    # ---- oracle replay, call by call
    o_oc5 = OS.split2octree_small(split_small.cpu(), d_small, CFG3['full_depth'])
    o_doc5 = OD.OracleDualOctree(o_oc5)
    o_doc5.post_processing_for_docnn()
    o_oc7 = OS.split2octree_large(o_oc5, res['hr'][n5 - nn5:].cpu(), d_small)
    o_doc7 = OD.OracleDualOctree(o_oc7)
    o_doc7.post_processing_for_docnn()
    assert o_doc5.total_num == n5 and o_doc7.total_num == n7
    onet = {'lr': lambda x, ls, xs: OU.lr_forward(parts['unet_lr'], st['lr'], x, ls, xs, None),
            'hr': lambda x, ls, xs: OU.hr_forward(parts['unet_hr'], st['hr'], x, o_doc5, ls, None, parts['unet_lr'], st['lr']),
            'feature': lambda x, ls, xs: OU.hr_forward(parts['unet_feature'], st['feature'], x, o_doc7, ls, None,
                                                       parts['unet_hr'], st['hr'])}
    times = OS.get_sampling_timesteps(B, steps)
    worst = {}
    final = {'lr': res['split_small'], 'hr': res['hr'], 'feature': res['feature']}
    for k, call in enumerate(rec.calls):
        stage, i = call['stage'], k % steps
        t, t_next = times[i]
        ls, lsn = OS.beta_linear_log_snr(t), OS.beta_linear_log_snr(t_next)
        assert torch.allclose(call['t'], ls, rtol=1e-6, atol=1e-6)                  # conditioning = log-SNR(t)
        if i == 0:
            assert torch.equal(call['x'], noises[stage]['init']) and call['xsc'] is None
        y_o = onet[stage](call['x'], ls, call['xsc'])
        worst[stage] = max(worst.get(stage, 0.0), rel(call['y'], y_o))
        # DDIM x0 update (octfusion_model_union.py:327-344) from the PRODUCT's output, sign/truncation as the reference
        out = call['y'].clone()
        trunc = 0.7 if stage == 'lr' else 0.0
        if bool(t[0] < trunc) and stage == 'lr':
            out.sign_()
        pl, pln = OS._pad(out, ls), OS._pad(out, lsn)
        alpha, sigma = OS.log_snr_to_alpha_sigma(pl)
        alpha_next, sigma_next = OS.log_snr_to_alpha_sigma(pln)
        c = -torch.special.expm1(pl - pln)
        mean = alpha_next * (call['x'] * (1 - c) / alpha + c * out)
        noise = torch.where(OS._pad(out, t_next > trunc), noises[stage]['steps'][i], torch.zeros_like(out))
        x_next = mean + torch.sqrt((sigma_next ** 2) * c) * noise
        got = rec.calls[k + 1]['x'] if i + 1 < steps else final[stage].cpu()
        assert rel(got, x_next) < 2e-5, (stage, i, rel(got, x_next))
        if i + 1 < steps:                                                           # self-conditioning = previous x_start
            nxt = rec.calls[k + 1]['xsc']
            if stage == 'lr':
                assert nxt is not None and rel(nxt, out) < 1e-6
    for stage, e in worst.items():
        assert e < tol, (stage, e)
    return worst


def test_two_stage_seeding_is_reproducible_and_reference_ordered():
    """CascadeSampler.sample(seed=, save_index=): seed + save_index before the lr loop, seed before the hr loop
    (octfusion_model_union.py:372,390): two calls give identical shapes; the hr initial noise does not depend on
    save_index once split_small is given."""
    from octfusion_amd import pipeline, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    cfg = dict(CFG3, unet_type=['lr', 'hr'], input_depth=[3, 5], image_size=[8, 32], input_channels=[8, 3],
               out_channels=[8, 3], model_channels=[16, 32], num_res_blocks=[[1, 1, 1], [1, 1, 0]],
               channel_mult=[[1, 2, 4], [1, 2, 4]], df_type=['x0', 'eps'])
    net = UNet3DModel(**{k: v for k, v in dict(cfg, stage_flag='hr').items() if k != 'df_type'})
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    cs = pipeline.CascadeSampler(net, cfg, None)
    a = cs.sample(1, ddim_steps=3, seed=11, save_index=4)
    b = cs.sample(1, ddim_steps=3, seed=11, save_index=4)
    c = cs.sample(1, ddim_steps=3, seed=11, save_index=5)
    assert torch.equal(a['split_small'], b['split_small']) and torch.equal(a['hr'], b['hr'])
    assert not torch.equal(a['split_small'], c['split_small'])
    d = cs.sample(1, ddim_steps=3, seed=11, save_index=9, split_small=a['split_small'])
    assert torch.equal(d['hr'], a['hr'])


def test_three_stage_checkpoint_layout_roundtrip():
    """octfusion_model_union_3t.py:219-229 (save) / :250-261 (load): df_unet_feature / ema_df_unet_feature travel with
    the other two stages; optimiser state of the training AdamW round-trips through 'opt'."""
    from octfusion_amd import checkpoint as CK, training as TR
    from octfusion_amd.graph_unet_union import UNet3DModel
    kw = {k: v for k, v in dict(CFG3, stage_flag='feature').items() if k != 'df_type'}
    net = UNet3DModel(**kw)
    sd = C.fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()])
    net.load_state_dict(sd)
    net = net.to(dev()).eval()
    opt = TR.AdamW(TR.trainable_parameters(net, 'feature'), lr=1e-3)
    assert opt.params and all(k.startswith('unet_feature.') for k in opt.params)
    k0 = next(iter(opt.state))
    opt.state[k0][0].fill_(0.25)
    opt.step_count = 17
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'df_steps-latest.pth')
        CK.save_ckpt(path, net, net, 99, stage_flag='feature', opt_state=opt.state_dict())
        raw = torch.load(path, map_location='cpu', weights_only=False)
        assert set(raw) == {'df_unet_lr', 'ema_df_unet_lr', 'df_unet_hr', 'ema_df_unet_hr', 'df_unet_feature',
                            'ema_df_unet_feature', 'opt', 'global_step'}
        fresh = UNet3DModel(**kw).to(dev()).eval()
        opt2 = TR.AdamW(TR.trainable_parameters(fresh, 'feature'), lr=1e-3)
        assert CK.load_ckpt(path, fresh, fresh, opt=opt2) == 99
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), fresh.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert opt2.step_count == 17 and float(opt2.state[k0][0].flatten()[0]) == 0.25
    # a 2-stage file still loads into a 2-stage model, ignoring nothing silently
    with tempfile.TemporaryDirectory() as tmp:
        p2 = os.path.join(tmp, 'df2.pth')
        CK.save_ckpt(p2, net, net, 5, stage_flag='hr')
        assert set(torch.load(p2, map_location='cpu', weights_only=False)) == {
            'df_unet_lr', 'ema_df_unet_lr', 'df_unet_hr', 'ema_df_unet_hr', 'opt', 'global_step'}


def test_stage_step_freezes_the_nested_net(golden):
    """the second-stage training step must leave the nested (first-stage) weights bit-identical
    (octfusion_model_union.py:127-142: requires_grad False + optimiser over the trainable parameters only)."""
    from octfusion_amd import graph_unet_union as U, training as TR
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    G = golden('g_unet')
    oc = split2octree_small(G['split_small'].to(dev()), 5, 3)
    doc = DualOctree(oc)
    h, l = C.TINY_HR_CFG, C.TINY_LR_CFG
    cfg = dict(stage_flag='hr', image_size=[8, 32], input_depth=[3, 5], unet_type=['lr', 'hr'], full_depth=3,
               input_channels=[8, 3], out_channels=[8, 3], model_channels=[l['model_channels'], h['model_channels']],
               num_res_blocks=[[1, 1, 1], h['num_res_blocks']], attention_resolutions=[2, 4],
               channel_mult=[l['channel_mult'], h['channel_mult']], num_heads=4, use_checkpoint=False, dims=3)
    net = U.UNet3DModel(**cfg)
    net.load_state_dict(C.fill_state_dict([(k, tuple(v.shape)) for k, v in net.state_dict().items()]))
    net = net.to(dev())
    before = {k: v.clone() for k, v in net.unet_lr.state_dict().items()}
    hr_before = {k: v.clone() for k, v in net.unet_hr.state_dict().items()}
    # even an optimiser built over EVERY parameter (the old call pattern) must not touch the frozen stage
    opt = TR.AdamW(dict(net.named_parameters()), lr=1e-2, weight_decay=0.1)
    codes = C.rand_input('freeze_codes', doc.total_num, 3).to(dev())
    TR.hr_stage_step(net, opt, codes, doc, 5)
    for k, v in net.unet_lr.state_dict().items():
        assert torch.equal(v, before[k]), 'nested lr parameter %s changed' % k
    assert any(not torch.equal(v, hr_before[k]) for k, v in net.unet_hr.state_dict().items())


def test_batched_three_stage_sampling_with_lazy_step_noise():
    """CascadeSampler.sample(shape_indices=) on the 3-stage model: the sparse x0 stages draw their per-step noise
    lazily (pipeline.StepNoise; the feature stage's 200 tensors would be gigabytes up front).  Same samples as with
    every tensor drawn up front, and a batch of two equals the two shapes sampled alone."""
    from octfusion_amd import pipeline, synthetic
    from octfusion_amd.graph_unet_union import UNet3DModel
    net = UNet3DModel(**{k: v for k, v in dict(CFG3, stage_flag='feature').items() if k != 'df_type'})
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    cs = pipeline.CascadeSampler(net, CFG3, None)
    steps, seed, idxs = 5, 21, [6, 2]
    lazy = cs.sample(2, ddim_steps=steps, seed=seed, shape_indices=idxs, use_graph=False)

    class Eager(list):
        def __init__(self, draw, n):
            super().__init__(draw() for _ in range(n))

        def finish(self):
            pass
    saved = pipeline.StepNoise
    pipeline.StepNoise = Eager
    try:
        eager = cs.sample(2, ddim_steps=steps, seed=seed, shape_indices=idxs, use_graph=False)
    finally:
        pipeline.StepNoise = saved
    for k in ('split_small', 'hr', 'feature'):
        assert torch.equal(lazy[k], eager[k]), k
    replay = cs.sample(2, ddim_steps=steps, seed=seed, shape_indices=idxs)          # hipGraph replay path
    assert torch.equal(replay['split_small'], lazy['split_small'])
    assert replay['feature'].shape == lazy['feature'].shape
