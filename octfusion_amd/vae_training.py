"""Training step of the GraphVAE -- the remainder of SURVEY 8f-4.

Mirror of reference models/octfusion_model_vae.py:178-188, 255-262 (forward -> geometry_loss -> backward -> AdamW):
GraphVAE.forward with a ground-truth output octree (graph_vae.py:246-289), the objectives of
dualoctree_networks/loss.py:164-178 (octree cross entropy per depth, MPU value / gradient regression with
'sdf_reg_loss', KL of the posterior) and their gradients w.r.t. every parameter.  The reference differentiates
through torch.autograd (twice through the MPU: loss.py:12-20 uses create_graph=True); here every op has an explicit
backward on libofx: ofx_mpu_eval_grad / ofx_mpu_backward, ofx_octree_ce, ofx_sdf_reg_loss, ofx_kl_sample_fwd/bwd
(csrc/ofx_loss.hip, csrc/ofx_graph.hip) and the GraphConv / GroupNorm / linear backward kernels of backward.py.
No CPU path.
"""
import ctypes

import torch

from . import backward as BW
from . import _lib, mpu, ops
from ._lib import call, ptr, stream
from .modules import pool_nodes, unpool_nodes


# ------------------------------------------------------------------------------------------------ loss kernels
def _f32(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise _lib.OfxError('VAE losses need fp32 HIP tensors (no CPU path)')
    return t.contiguous()


def mpu_eval_grad(octree, depth_start, depth_end, pos, reg):
    """(sdf [n], grad [n,3], mask [n]) -- get_linear_pred + compute_gradient (mpu.py:97-134, loss.py:12-20)."""
    h = mpu._TreeHandle.of(octree)
    reg = mpu._code(reg, h, depth_start, depth_end)
    pos = _f32(pos)
    n = pos.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=pos.device)
    grad = torch.empty(n, 3, dtype=torch.float32, device=pos.device)
    mask = torch.empty(n, dtype=torch.uint8, device=pos.device)
    call('ofx_mpu_eval_grad', ctypes.byref(h.tree), depth_start, depth_end, ptr(pos), n, ptr(reg), ptr(sdf), ptr(grad),
         ptr(mask), stream())
    return sdf, grad, mask.bool()


def mpu_backward(octree, depth_start, depth_end, pos, reg, dsdf, dgrad):
    """dL/dreg [rows,4] from dL/dsdf [n] and dL/dgrad [n,3] (the double backward of loss.py:12-20)."""
    h = mpu._TreeHandle.of(octree)
    reg = mpu._code(reg, h, depth_start, depth_end)
    pos = _f32(pos)
    dreg = torch.zeros_like(reg)
    call('ofx_mpu_backward', ctypes.byref(h.tree), depth_start, depth_end, ptr(pos), pos.shape[0], ptr(reg),
         ptr(_f32(dsdf)) if dsdf is not None else None, ptr(_f32(dgrad)) if dgrad is not None else None, ptr(dreg),
         stream())
    return dreg


def octree_ce(logits, child, weight=1.0, need_grad=True):
    """compute_octree_loss for one depth (loss.py:110-122): (loss, accuracy, dL/dlogits) with label = child >= 0;
    loss / accuracy are 0-d device tensors (no host sync)."""
    logits = _f32(logits)
    n = logits.shape[0]
    assert logits.shape[1] == 2 and child.dtype == torch.int32 and child.shape[0] == n
    sums = torch.zeros(2, dtype=torch.float64, device=logits.device)
    dl = torch.empty_like(logits) if need_grad else None
    call('ofx_octree_ce', ptr(logits), 2, ptr(child), n, weight / max(n, 1), ptr(sums), ptr(dl) if need_grad else None, 2,
         stream())
    return (sums[0] * (weight / max(n, 1))).float(), (sums[1] / max(n, 1)).float(), dl


def sdf_reg_loss(sdf, grad, sdf_gt, grad_gt, w_sdf=200.0, w_grad=1.0, need_grad=True):
    """sdf_reg_loss (loss.py:23-29): (grad_loss, sdf_loss, dL/dsdf, dL/dgrad)."""
    sdf, grad, sdf_gt, grad_gt = _f32(sdf), _f32(grad), _f32(sdf_gt), _f32(grad_gt)
    n = sdf.shape[0]
    assert grad.shape == (n, 3) and grad_gt.shape == (n, 3) and sdf_gt.shape == (n,)
    sums = torch.zeros(2, dtype=torch.float64, device=sdf.device)
    dsdf = torch.empty_like(sdf) if need_grad else None
    dgrad = torch.empty_like(grad) if need_grad else None
    call('ofx_sdf_reg_loss', ptr(sdf), ptr(grad), ptr(sdf_gt), ptr(grad_gt), n, w_sdf, w_grad, ptr(sums),
         ptr(dsdf) if need_grad else None, ptr(dgrad) if need_grad else None, stream())
    return (sums[0] * (w_grad / max(3 * n, 1))).float(), (sums[1] * (w_sdf / max(n, 1))).float(), dsdf, dgrad


def kl_sample(params, noise, embed_dim):
    """DiagonalGaussianDistribution(params).sample() with the given noise and kl().mean() (distributions.py:24-46)."""
    params = _f32(params)
    n = params.shape[0]
    z = torch.empty(n, embed_dim, dtype=torch.float32, device=params.device)
    kl = torch.zeros(1, dtype=torch.float64, device=params.device)
    call('ofx_kl_sample_fwd', ptr(params), params.shape[1], ptr(_f32(noise)) if noise is not None else None, n,
         embed_dim, ptr(z), ptr(kl), stream())
    return z, (kl[0] / max(n * embed_dim, 1)).float()


def kl_sample_backward(params, noise, dz, embed_dim, kl_weight):
    params = _f32(params)
    n = params.shape[0]
    dp = torch.empty_like(params)
    call('ofx_kl_sample_bwd', ptr(params), params.shape[1], ptr(_f32(noise)) if noise is not None else None,
         ptr(_f32(dz)), n, embed_dim, kl_weight / max(n * embed_dim, 1), ptr(dp), dp.shape[1], stream())
    return dp


# ------------------------------------------------------------------------------------- module forward / backward
def _vres_fwd(blk, x, doctree, d):
    """VAE GraphResBlock (modules.py:593-641) keeping what the backward needs."""
    h1 = blk.norm1(x, doctree, d, act='silu')
    c1 = blk.conv1(h1, doctree, d)
    h2 = blk.norm2(c1, doctree, d, act='silu')
    lin = None
    skip = x
    if blk.channel_in != blk.channel_out:
        lin = blk.conv1x1c.conv(x)
        skip = blk.conv1x1c.gn(lin, doctree, d)
    return blk.conv2(h2, doctree, d, res=skip), (x, h1, c1, h2, lin)


def _vres_bwd(blk, saved, dy, doctree, d, G, prefix):
    x, h1, c1, h2, lin = saved
    dh2 = BW._gconv_bwd(blk.conv2, h2, dy, doctree, d, G, prefix + 'conv2.')
    dc1 = BW._dgn_bwd(blk.norm2, c1, dh2, doctree, d, 'silu', G, prefix + 'norm2.')
    dh1 = BW._gconv_bwd(blk.conv1, h1, dc1, doctree, d, G, prefix + 'conv1.')
    dx = BW._dgn_bwd(blk.norm1, x, dh1, doctree, d, 'silu', G, prefix + 'norm1.')
    if lin is None:
        dx += dy
    else:
        dlin = BW._dgn_bwd(blk.conv1x1c.gn, lin, dy, doctree, d, None, G, prefix + 'conv1x1c.gn.')
        dx += BW._linear_bwd(blk.conv1x1c.conv.linear, x, dlin, G, prefix + 'conv1x1c.conv.linear.')
    return dx


def _blocks_fwd(blocks, x, doctree, d):
    tape = []
    for blk in blocks.resblks:
        x, s = _vres_fwd(blk, x, doctree, d)
        tape.append(s)
    return x, tape


def _blocks_bwd(blocks, tape, dy, doctree, d, G, prefix):
    for i in range(len(tape) - 1, -1, -1):
        dy = _vres_bwd(blocks.resblks[i], tape[i], dy, doctree, d, G, prefix + 'resblks.%d.' % i)
    return dy


def _cgg_fwd(m, x, doctree, d):
    """Conv1x1GnGelu (modules.py:353-365)."""
    lin = m.conv(x)
    return m.gn(lin, doctree, d, act='gelu'), (x, lin)


def _cgg_bwd(m, saved, dy, doctree, d, G, prefix):
    x, lin = saved
    dlin = BW._dgn_bwd(m.gn, lin, dy, doctree, d, 'gelu', G, prefix + 'gn.')
    return BW._linear_bwd(m.conv.linear, x, dlin, G, prefix + 'conv.linear.')


def _head_fwd(head, x, doctree, d):
    """_make_predict_module (graph_vae.py:127-130): Conv1x1GnGelu -> Conv1x1(bias)."""
    h, s = _cgg_fwd(head[0], x, doctree, d)
    return head[1](h), (s, h)


def _head_bwd(head, saved, dy, doctree, d, G, prefix):
    s, h = saved
    dh = BW._linear_bwd(head[1].linear, h, dy, G, prefix + '1.linear.')
    return _cgg_bwd(head[0], s, dh, doctree, d, G, prefix + '0.')


@torch.no_grad()
def vae_forward_backward(vae, data, doctree_in, doctree_out, pos, sdf_gt, grad_gt, noise=None, kl_weight=1.0):
    """GraphVAE.forward(octree_in, octree_gt, pos) + geometry_loss('sdf_reg_loss') + backward.
    data [N_depth, channel_in]: the input feature on doctree_in's finest graph; doctree_out: dual octree of the
    ground-truth octree; pos [n,4], sdf_gt [n], grad_gt [n,3].  Returns (losses {name: 0-d tensor} incl. 'loss' =
    the sum octfusion_model_vae.py:182-183 optimises, outputs {'logits','reg_voxs','mpus','z'},
    {state_dict key: gradient})."""
    G = BW._Grads()
    depth, ds, dout, fd = vae.depth, vae.depth_stop, vae.depth_out, vae.full_depth
    E = vae.KL_conv.linear.weight.shape[0] // 2
    # ------------------------------------------------------------------ encoder (graph_vae.py:134-170)
    enc_tape = []
    convd = data
    for i, d in enumerate(range(depth, ds - 1, -1)):
        s_in = None
        if d == depth:
            s_in = convd
            convd = vae.conv1(convd, doctree_in, d)
        convd, t_blk = _blocks_fwd(vae.encoder[i], convd, doctree_in, d)
        s_down = None
        if d > ds:
            x_d = convd
            p = pool_nodes(x_d, doctree_in, d, vae.downsample[i].downsample)
            if vae.downsample[i].channels_in != vae.downsample[i].channels_out:
                convd, s_c = _cgg_fwd(vae.downsample[i].conv1x1, p, doctree_in, d - 1)
            else:
                convd, s_c = p, None
            s_down = (x_d, s_c)
        enc_tape.append((d, s_in, t_blk, s_down))
    h_pre = convd
    h = vae.encoder_norm_out(h_pre, doctree_in, ds, act='gelu')
    params = vae.KL_conv(h)
    if noise is None:
        noise = torch.randn(params.shape[0], E, device=params.device)
    z, kl_mean = kl_sample(params, noise, E)
    # ------------------------------------------------------------------ decoder (graph_vae.py:171-223)
    x0 = vae.post_KL_conv(z)
    x1, t_m1 = _blocks_fwd(vae.decoder_mid.block_1, x0, doctree_out, ds)
    x2, t_m2 = _blocks_fwd(vae.decoder_mid.block_2, x1, doctree_out, ds)
    octree_out = doctree_out.octree
    deconv = x2
    dec_tape = []
    logits, reg_voxs, mpus, losses = {}, {}, {}, {}
    d_heads = {}
    for i, d in enumerate(range(ds, dout + 1)):
        s_up = None
        if d > ds:
            up = vae.upsample[i - 1]
            x_u = deconv
            u = unpool_nodes(x_u, doctree_out, d - 1, up.upsample)
            if up.channels_in != up.channels_out:
                deconv, s_c = _cgg_fwd(up.conv1x1, u, doctree_out, d)
            else:
                deconv, s_c = u, None
            s_up = (x_u, s_c)
        deconv, t_blk = _blocks_fwd(vae.decoder[i], deconv, doctree_out, d)
        logit, s_pred = _head_fwd(vae.predict[i], deconv, doctree_out, d)
        reg, s_reg = _head_fwd(vae.regress[i], deconv, doctree_out, d)
        nnum = int(doctree_out.nnum[d])
        n_d = logit.shape[0]
        logits[d] = logit[n_d - nnum:]
        dmap = doctree_out.pad_rows(d)
        pad = torch.zeros(doctree_out.graph[d]['node_mask'].shape[0], reg.shape[1], dtype=torch.float32, device=reg.device)
        ops.rows_copy(reg, pad, n_d, dmap=dmap)
        reg_voxs[d] = pad
        # ---- objectives of this depth and their gradients w.r.t. the two heads' outputs
        losses['loss_%d' % d], losses['accu_%d' % d], dl_tail = octree_ce(logits[d], octree_out.children[d])
        dlogit = torch.zeros_like(logit)
        dlogit[n_d - nnum:] = dl_tail
        sdf, grad, mask = mpu_eval_grad(octree_out, fd, d, pos, pad)
        mpus[d] = (sdf, mask)
        losses['grad_loss_%d' % d], losses['sdf_loss_%d' % d], dsdf, dgrad = sdf_reg_loss(sdf, grad, sdf_gt, grad_gt)
        dpad = mpu_backward(octree_out, fd, d, pos, pad, dsdf, dgrad)
        dreg = torch.empty_like(reg)
        ops.rows_copy(dpad, dreg, n_d, smap=dmap)
        d_heads[d] = (dlogit, dreg)
        dec_tape.append((d, s_up, t_blk, deconv, s_pred, s_reg))
    losses['kl_loss'] = kl_mean * kl_weight
    losses['loss'] = torch.stack([v for k, v in losses.items() if 'loss' in k]).sum()
    # ================================================================================================= backward
    dnext = None
    for i in range(len(dec_tape) - 1, -1, -1):
        d, s_up, t_blk, deconv_d, s_pred, s_reg = dec_tape[i]
        dlogit, dreg = d_heads[d]
        dd = _head_bwd(vae.regress[i], s_reg, dreg, doctree_out, d, G, 'regress.%d.' % i)
        dd += _head_bwd(vae.predict[i], s_pred, dlogit, doctree_out, d, G, 'predict.%d.' % i)
        if dnext is not None:
            dd += dnext
        dx = _blocks_bwd(vae.decoder[i], t_blk, dd, doctree_out, d, G, 'decoder.%d.' % i)
        if s_up is not None:
            up = vae.upsample[i - 1]
            x_u, s_c = s_up
            if s_c is not None:
                dx = _cgg_bwd(up.conv1x1, s_c, dx, doctree_out, d, G, 'upsample.%d.conv1x1.' % (i - 1))
            dnext = BW._unpool_bwd(up.upsample, x_u, dx, doctree_out, d - 1, G, 'upsample.%d.upsample.' % (i - 1))
        else:
            dnext = dx
    dx = _blocks_bwd(vae.decoder_mid.block_2, t_m2, dnext, doctree_out, ds, G, 'decoder_mid.block_2.')
    dx = _blocks_bwd(vae.decoder_mid.block_1, t_m1, dx, doctree_out, ds, G, 'decoder_mid.block_1.')
    dz = BW._linear_bwd(vae.post_KL_conv.linear, z, dx, G, 'post_KL_conv.linear.')
    dparams = kl_sample_backward(params, noise, dz, E, kl_weight)
    dh = BW._linear_bwd(vae.KL_conv.linear, h, dparams, G, 'KL_conv.linear.')
    dconv = BW._dgn_bwd(vae.encoder_norm_out, h_pre, dh, doctree_in, ds, 'gelu', G, 'encoder_norm_out.')
    for i in range(len(enc_tape) - 1, -1, -1):
        d, s_in, t_blk, s_down = enc_tape[i]
        if s_down is not None:
            x_d, s_c = s_down
            down = vae.downsample[i]
            if s_c is not None:
                dconv = _cgg_bwd(down.conv1x1, s_c, dconv, doctree_in, d - 1, G, 'downsample.%d.conv1x1.' % i)
            dconv = BW._pool_bwd(down.downsample, x_d, dconv, doctree_in, d, G, 'downsample.%d.downsample.' % i)
        dconv = _blocks_bwd(vae.encoder[i], t_blk, dconv, doctree_in, d, G, 'encoder.%d.' % i)
        if s_in is not None:
            BW._gconv_bwd(vae.conv1, s_in, dconv, doctree_in, d, G, 'conv1.', need_dx=False)
    out = {'logits': logits, 'reg_voxs': reg_voxs, 'mpus': mpus, 'z': z, 'octree_out': octree_out}
    return losses, out, dict(G)


def poly_lr(base_lr, epoch, epochs, power=0.9):
    """octfusion_model_vae.py:93-96: LambdaLR factor (1 - epoch / epochs) ** 0.9."""
    return base_lr * (1.0 - epoch / epochs) ** power


@torch.no_grad()
def vae_stage_step(vae, opt, data, doctree_in, doctree_out, pos, sdf_gt, grad_gt, noise=None, kl_weight=1.0):
    """One optimisation step of the VAE (octfusion_model_vae.py:255-262).  opt: training.AdamW over
    vae.named_parameters().  Returns the dict of losses (0-d device tensors)."""
    losses, _, grads = vae_forward_backward(vae, data, doctree_in, doctree_out, pos, sdf_gt, grad_gt, noise, kl_weight)
    opt.step(grads)
    return losses
