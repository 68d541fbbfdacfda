"""Oracle: the two U-Nets as pure functions over a reference state_dict.  TEST INFRASTRUCTURE.

``hr_forward`` restates reference
models/networks/diffusion_networks/graph_unet_hr.py:214-281 (ctor :69-209 fixes
the block order, reproduced by :func:`hr_plan`); ``lr_forward`` /
``lr_forward_as_middle`` restate graph_unet_lr.py:175-230.  State-dict keys are
the reference's, so one checkpoint feeds the reference
(golden generation), this oracle and the product modules.
"""
import math

import torch
import torch.nn.functional as F

from . import modules as M
from .octree import octree2voxel


def timestep_embedding(timesteps, dim, max_period=10000):
    """ldm_diffusion_util.py:171-191."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=M.FLOAT) / half)
    args = timesteps[:, None].to(M.FLOAT) * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def hr_plan(cfg):
    """Block order fixed by graph_unet_hr.py:116-209.

    Returns (input_plan, output_plan, d_mid): lists of (kind, n_node_type) where
    kind in {'conv','res','down','up'}; n_node_type is the ctor value (depth-1).
    """
    d = cfg['input_depth']
    mult = cfg['channel_mult']
    nres = cfg['num_res_blocks']
    inp = [('conv', d - 1)]
    for level in range(len(mult)):
        for _ in range(nres[level]):
            inp.append(('res', d - 1))
        if level != len(mult) - 1:
            d -= 1
            inp.append(('down', d - 1))
    d_mid = d
    out = []
    for level in reversed(range(len(mult))):
        for i in range(nres[level] + 1):
            out.append(('res', d - 1))
            if level and i == nres[level]:
                d += 1
                out.append(('up', d - 1))
    return inp, out, d_mid


def hr_forward(sd, cfg, x, doctree, timesteps, label=None, lr_sd=None, lr_cfg=None,
               as_middle=False):
    """graph_unet_hr.py:214-281.  ``lr_sd`` is the nested stage's state_dict (or None)."""
    mc = cfg['model_channels']
    emb = timestep_embedding(timesteps, mc)
    emb = F.linear(emb, sd['time_embed.0.weight'], sd['time_embed.0.bias'])
    emb = F.linear(M.silu(emb), sd['time_embed.2.weight'], sd['time_embed.2.bias'])
    if cfg.get('num_classes') is not None:
        assert label is not None and label.shape == (doctree.batch_size,)
        emb = emb + sd['label_emb.weight'][label]
    else:
        assert label is None

    inp, outp, _ = hr_plan(cfg)
    d = cfg['input_depth']
    hs = []
    if not as_middle:
        h = M.graph_conv(x, doctree, d, sd['input_blocks.0.weights'], None, inp[0][1])
    else:
        h = x
    hs.append(h)
    for i, (kind, nt) in enumerate(inp):
        if i == 0:
            continue
        blk = M._sub(sd, 'input_blocks.%d' % i)
        if kind == 'res':
            h = M.graph_resblock_embed(h, emb, doctree, d, blk, nt)
        elif kind == 'down':
            h = M.graph_downsample(h, doctree, d, blk, nt)
            d -= 1
        hs.append(h)

    if lr_sd is not None:
        h = M.graph_resblock_embed(h, emb, doctree, d, M._sub(sd, 'middle_block1'), d - 1)
        if lr_cfg.get('kind', 'lr') == 'lr':
            h_lr = lr_forward_as_middle(lr_sd, lr_cfg, h, doctree, timesteps, label)
        else:   # 3-stage: the nested net is another hr net run as_middle without its own lr
            h_lr = hr_forward(lr_sd, lr_cfg, h, doctree, timesteps, label, None, None, True)
        h = torch.cat([h, h_lr], dim=1)
        h = M.graph_resblock_embed(h, emb, doctree, d, M._sub(sd, 'middle_block2'), d - 1)

    for i, (kind, nt) in enumerate(outp):
        blk = M._sub(sd, 'output_blocks.%d' % i)
        if kind == 'res':
            h = torch.cat([h, hs.pop()], dim=1)
            h = M.graph_resblock_embed(h, emb, doctree, d, blk, nt)
        else:
            h = M.graph_upsample(h, doctree, d, blk, nt)
            d += 1

    h = M.silu(M.dual_octree_group_norm(h, doctree, d, sd['end_norm.weights'], sd['end_norm.bias']))
    if as_middle:
        return h
    out = M.graph_conv(h, doctree, d, sd['out.weights'], None, cfg['input_depth'] - 1)
    assert out.shape[0] == x.shape[0]
    return out


def lr_forward(sd, cfg, x, timesteps, x_self_cond=None, label=None, as_middle=False):
    """graph_unet_lr.py:184-230."""
    nh = cfg['num_heads']
    mc = cfg['model_channels']
    mult = cfg['channel_mult']
    attn_res = cfg['attention_resolutions']
    if not as_middle:
        if x_self_cond is None:
            x_self_cond = torch.zeros_like(x)
        x = torch.cat((x, x_self_cond), dim=1)
        x = F.conv3d(x, sd['input_emb.weight'], sd['input_emb.bias'], padding=1)

    # LearnedSinusoidalPosEmb modules.py:557-563 + time_emb graph_unet_lr.py:107-111
    t = timesteps[:, None]
    freqs = t * sd['time_pos_emb.weights'][None, :] * 2 * math.pi
    four = torch.cat((t, freqs.sin(), freqs.cos()), dim=-1)
    emb = F.linear(four, sd['time_emb.0.weight'], sd['time_emb.0.bias'])
    emb = F.linear(M.silu(emb), sd['time_emb.2.weight'], sd['time_emb.2.bias'])
    if cfg.get('num_classes') is not None:
        assert label.shape == (x.shape[0],)
        emb = emb + sd['label_emb.weight'][label]

    nlev = len(mult)
    hs = []
    ds = 1
    for ind in range(nlev):
        last = ind >= nlev - 1
        x = M.resnet_block(x, emb, M._sub(sd, 'downs.%d.0' % ind))
        if ds in attn_res:
            x = M.attn_seq(x, M._sub(sd, 'downs.%d.1' % ind), nh)
        hs.append(x)
        if not last:
            x = F.conv3d(x, sd['downs.%d.2.op.weight' % ind], sd['downs.%d.2.op.bias' % ind],
                         stride=2, padding=1)
            ds *= 2
    x = M.resnet_block(x, emb, M._sub(sd, 'mid_block1'))
    if ds in attn_res:
        x = M.attn_seq(x, M._sub(sd, 'mid_self_attn'), nh)
    x = M.resnet_block(x, emb, M._sub(sd, 'mid_block2'))
    # graph_unet_lr.py:152-166: `ups` walks reversed(in_out[1:]) and is_last is never true
    for ind in range(nlev - 1):
        x = torch.cat((x, hs.pop()), dim=1)
        x = M.resnet_block(x, emb, M._sub(sd, 'ups.%d.0' % ind))
        if ds in attn_res:
            x = M.attn_seq(x, M._sub(sd, 'ups.%d.1' % ind), nh)
        x = F.interpolate(x, scale_factor=2, mode='nearest')
        x = F.conv3d(x, sd['ups.%d.2.conv.weight' % ind], sd['ups.%d.2.conv.bias' % ind], padding=1)
        ds //= 2
    x = M.silu(M.group_norm32(x, sd['end.0.weight'], sd['end.0.bias'], mc))
    if as_middle:
        return x
    return F.conv3d(x, sd['out.weight'], sd['out.bias'], padding=1)


def lr_forward_as_middle(sd, cfg, h, doctree, timesteps, label):
    """graph_unet_lr.py:175-182: octree2voxel -> dense U-Net -> gather back."""
    fd = cfg['full_depth']
    vox = octree2voxel(h, doctree.octree, fd).permute(0, 4, 1, 2, 3).contiguous()
    out = lr_forward(sd, cfg, vox, timesteps, None, label, as_middle=True)
    x, y, z, b = doctree.octree.xyzb(fd)
    return out.permute(0, 2, 3, 4, 1).contiguous()[b, x, y, z, :]
