"""GPU parity of the VAE training step (SURVEY 8f-4 remainder): loss kernels, NeuralMPU gradient / adjoint and the
whole forward + backward against (a) what the reference's own GraphVAE.forward + loss.geometry_loss + autograd
produced (tests/golden/g_vae_train.pt) and (b) autograd through the CPU oracle on larger seeded inputs.
Tolerances: losses 5e-4 relative; gradients 5e-3 relative to the tensor's range (bf16x3 contractions, fp32
atomics in the adjoint); masks / labels / accuracies exact."""
import pytest
import torch

import common as C
from test_gpu_parity import close, dev, tiny

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def sub_close(actual, rec, rel=5e-3):
    f = actual.detach().float().cpu().reshape(-1)
    assert tuple(actual.shape) == tuple(rec['shape']), (tuple(actual.shape), rec['shape'])
    scale = max(float(rec['vals'].abs().max()), 1e-30)
    err = float((f[::rec['stride']] - rec['vals']).abs().max()) / scale
    assert err < rel, 'rel-to-max error %.3e' % err
    assert abs(float(f.double().norm()) - rec['norm']) <= rel * rec['norm'] + 1e-12


def load_vae(G):
    from octfusion_amd.graph_vae import GraphVAE
    vae = GraphVAE(**G['cfg'])
    vae.load_state_dict(C.fill_state_dict(G['keys']), strict=True)
    return vae.to(dev())


def train_case(G):
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_large
    oc, _ = tiny(G['split_small'])
    oc_l = split2octree_large(oc, G['split_large'].to(dev()), 4)
    doc_l = DualOctree(oc_l)
    data = C.rand_input('vae_enc_in', doc_l.csr(6)[2], 4).to(dev())
    noise = C.rand_input('vae_post_noise', *G['n_noise']).to(dev())
    return oc_l, doc_l, data, noise


def test_vae_training_step_golden(golden):
    """forward + losses + every parameter gradient against the reference's own run."""
    from octfusion_amd import vae_training as VT
    G = golden('g_vae_train')
    vae = load_vae(G)
    oc_l, doc_l, data, noise = train_case(G)
    losses, out, grads = VT.vae_forward_backward(vae, data, doc_l, doc_l, G['pos'].to(dev()), G['sdf_gt'].to(dev()),
                                                 G['grad_gt'].to(dev()), noise, G['kl_weight'])
    for k, v in G['losses'].items():
        got = float(losses[k])
        tol = 1e-3 if k.startswith('accu') else 5e-4 * abs(v) + 1e-6        # accuracy: a near-tie argmax may flip
        assert abs(got - v) <= tol, (k, got, v)
    assert abs(float(losses['loss']) - G['total']) <= 5e-4 * G['total']
    for d in (4, 5, 6):
        sub_close(out['logits'][d], G['logits'][d], 1e-3)
        sub_close(out['reg_voxs'][d], G['reg_voxs'][d], 1e-3)
        close(out['mpus'][d][0], G['sdf'][d], 1e-3)
    n = 0
    params = dict(vae.named_parameters())
    for k, rec in G['grads'].items():
        if rec is None:
            assert k not in grads or float(grads[k].abs().max()) == 0.0, k
            continue
        assert k in grads, k
        assert grads[k].shape == params[k].shape, k
        sub_close(grads[k], rec)
        n += 1
    assert n > 100 and set(grads) <= set(params)


def test_mpu_gradient_and_adjoint_vs_oracle(golden):
    """ofx_mpu_eval_grad / ofx_mpu_backward against autograd through oracle/mpu.py on the g_mpu tree: 4096 points
    incl. exact cell-centre planes (the |.| kink) and the cube boundary; random upstream gradients."""
    from oracle import loss as OL
    from oracle import mpu as OMPU
    from oracle import sampler as OS
    from octfusion_amd import vae_training as VT
    from octfusion_amd.octree import split2octree_large
    G = golden('g_mpu')
    oc, _ = tiny(G['split_small'])
    oc_l = split2octree_large(oc, G['split_large'].to(dev()), 4)
    o_oc = OS.split2octree_large(OS.split2octree_small(G['split_small'], 4, 2), G['split_large'], 4)
    fd, ds, dp = G['cfg']
    ncum = torch.cumsum(oc_l.nnum, 0)
    pos = G['pos']
    n = pos.shape[0]
    up_s = C.rand_input('mpu_up_sdf', n, 1).view(-1)
    up_g = C.rand_input('mpu_up_grad', n, 3)
    for d in range(ds, dp + 1):
        reg = C.rand_input('mpu_code_%d' % d, int(ncum[d] - (ncum[fd - 1] if fd else 0)), 4)
        with torch.enable_grad():
            p = pos.clone().requires_grad_(True)
            r = reg.clone().requires_grad_(True)
            sdf_o, mask_o = OMPU.linear_pred(p, o_oc, r, fd, d)
            g_o = OL.compute_gradient(sdf_o, p)[:, :3]
            ((sdf_o * up_s).sum() + (g_o * up_g).sum()).backward()
        sdf, grad, mask = VT.mpu_eval_grad(oc_l, fd, d, pos.to(dev()), reg.to(dev()))
        assert torch.equal(mask.cpu(), mask_o)
        torch.testing.assert_close(sdf.cpu(), sdf_o.detach(), rtol=1e-4, atol=1e-5)
        close(grad, g_o.detach(), 1e-4)
        dreg = VT.mpu_backward(oc_l, fd, d, pos.to(dev()), reg.to(dev()), up_s.to(dev()), up_g.to(dev()))
        close(dreg, r.grad, 1e-4)
        # value-only and gradient-only upstreams are the two halves of the same adjoint
        d1 = VT.mpu_backward(oc_l, fd, d, pos.to(dev()), reg.to(dev()), up_s.to(dev()), None)
        d2 = VT.mpu_backward(oc_l, fd, d, pos.to(dev()), reg.to(dev()), None, up_g.to(dev()))
        close(d1 + d2, r.grad, 1e-4)


def test_loss_kernels_vs_oracle():
    """ofx_octree_ce / ofx_sdf_reg_loss / ofx_kl_sample_* against torch (F.cross_entropy, the formulas of
    loss.py:23-29, distributions.py:24-46) incl. ties, huge logits, clamped log-variances and empty inputs."""
    import torch.nn.functional as F
    from oracle import loss as OL
    from octfusion_amd import vae_training as VT
    g = torch.Generator().manual_seed(5)
    n = 100_003
    logits = torch.randn(n, 2, generator=g) * 3
    logits[:10] = torch.tensor([[80.0, -80.0]] * 5 + [[0.5, 0.5]] * 5)              # saturated rows, exact ties
    child = torch.where(torch.rand(n, generator=g) < 0.4, torch.arange(n), torch.full((n,), -1)).to(torch.int32)
    with torch.enable_grad():
        lg = logits.clone().requires_grad_(True)
        label = (child >= 0).long()
        ref = F.cross_entropy(lg, label)
        ref.backward()
    loss, accu, dl = VT.octree_ce(logits.to(dev()), child.to(dev()))
    assert abs(float(loss) - float(ref)) <= 1e-5 * float(ref)
    assert float(accu) == float(logits.argmax(1).eq(label).float().mean())
    close(dl, lg.grad, 1e-5)
    # sdf_reg_loss
    sdf, sg = torch.randn(n, generator=g), torch.randn(n, generator=g)
    grad, gg = torch.randn(n, 3, generator=g), torch.randn(n, 3, generator=g)
    with torch.enable_grad():
        s, gr = sdf.clone().requires_grad_(True), grad.clone().requires_grad_(True)
        o = OL.sdf_reg_loss(s, gr, sg, gg)
        (o['grad_loss'] + o['sdf_loss']).backward()
    gl, sl, dsdf, dgrad = VT.sdf_reg_loss(sdf.to(dev()), grad.to(dev()), sg.to(dev()), gg.to(dev()))
    assert abs(float(gl) - float(o['grad_loss'])) <= 1e-5 * float(o['grad_loss'])
    assert abs(float(sl) - float(o['sdf_loss'])) <= 1e-5 * float(o['sdf_loss'])
    close(dsdf, s.grad, 1e-5)
    close(dgrad, gr.grad, 1e-5)
    # posterior
    E = 3
    params = torch.randn(n, 2 * E, generator=g)
    params[:4, E:] = torch.tensor([[-40.0, 25.0, -30.0], [20.0, 0.0, 1.0], [-31.0, 21.0, 3.0], [0.0, 0.0, 0.0]])
    noise = torch.randn(n, E, generator=g)
    dz = torch.randn(n, E, generator=g)
    with torch.enable_grad():
        pr = params.clone().requires_grad_(True)
        z_o, kl_o = OL.posterior(pr, noise)
        ((z_o * dz).sum() + 0.1 * kl_o.mean()).backward()
    z, kl = VT.kl_sample(params.to(dev()), noise.to(dev()), E)
    close(z, z_o.detach(), 1e-6)
    assert abs(float(kl) - float(kl_o.mean())) <= 1e-5 * float(kl_o.mean())
    dp = VT.kl_sample_backward(params.to(dev()), noise.to(dev()), dz.to(dev()), E, 0.1)
    close(dp, pr.grad, 1e-5)
    # empty inputs are accepted and yield zero sums
    l0, a0, d0 = VT.octree_ce(torch.empty(0, 2, device=dev()), torch.empty(0, dtype=torch.int32, device=dev()))
    assert float(l0) == 0.0 and d0.shape == (0, 2)


def test_vae_stage_step_updates_parameters(golden):
    """vae_stage_step = forward/backward + AdamW on every parameter: first update equals torch.optim.AdamW's on
    the same gradient, and a few steps on a fixed batch reduce the objective."""
    from octfusion_amd import training as T
    from octfusion_amd import vae_training as VT
    G = golden('g_vae_train')
    vae = load_vae(G)
    oc_l, doc_l, data, noise = train_case(G)
    pos, sg, gg = G['pos'].to(dev()), G['sdf_gt'].to(dev()), G['grad_gt'].to(dev())
    _, _, grads = VT.vae_forward_backward(vae, data, doc_l, doc_l, pos, sg, gg, noise, 0.1)
    name = 'regress.2.1.linear.weight'
    p0 = dict(vae.named_parameters())[name].detach().cpu().clone()
    ref_p = torch.nn.Parameter(p0.clone())
    lr = 1e-5          # small enough that the first-order decrease (lr * sum |g|) dominates on this random-weight net
    ref_opt = torch.optim.AdamW([ref_p], lr=lr)
    ref_p.grad = grads[name].cpu().clone()
    ref_opt.step()
    opt = T.AdamW(vae.named_parameters(), lr=lr)
    first = VT.vae_stage_step(vae, opt, data, doc_l, doc_l, pos, sg, gg, noise, 0.1)
    close(dict(vae.named_parameters())[name].detach().cpu() - p0, ref_p.detach() - p0, 1e-3)      # the update itself
    last = first
    for _ in range(4):
        last = VT.vae_stage_step(vae, opt, data, doc_l, doc_l, pos, sg, gg, noise, 0.1)
    assert float(last['loss']) < float(first['loss'])
    assert all(torch.isfinite(v).all() for v in last.values())
