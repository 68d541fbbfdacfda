"""Dense 16^3 voxel U-Net -- stage "lr", also nested as the middle of "hr".

Mirror of reference models/networks/diffusion_networks/graph_unet_lr.py
(``UNet3DModel``: ctor :65-173, forward :184-230, forward_as_middle :175-182)
with the reference's dense blocks (modules.py:26-95, 474-563): same constructor
keywords, signatures and state_dict keys (time_pos_emb.weights, time_emb.*,
input_emb.*, downs/ups.*, mid_*, end.*, out.*).

This part of the step is tiny and dense (B x 16^3 voxels; <5 % of the step's
FLOPs): 3x3x3 convolutions and GroupNorm use the
ROCm libraries through ATen (MIOpen / rocBLAS); the octree<->voxel permutations
at the boundary are libofx kernels.  No CPU path: forward_as_middle goes through
libofx and raises without it.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class GroupNorm32(nn.GroupNorm):
    def forward(self, x):
        return super().forward(x.float()).type(x.dtype)


def convnormalization(channels):
    return GroupNorm32(min(channels, 32), channels)


class our_Identity(nn.Module):
    def forward(self, x, *args, **kwargs):
        return x


class ConvUpsample(nn.Module):
    def __init__(self, channels, use_conv=True, dims=3):
        super().__init__()
        self.channels = channels
        self.conv = nn.Conv3d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='nearest'))


class ConvDownsample(nn.Module):
    def __init__(self, channels, use_conv=True, dims=3):
        super().__init__()
        self.channels = channels
        self.op = nn.Conv3d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.op(x)


class ResnetBlock(nn.Module):
    """reference modules.py:474-513 (use_text_condition False on every live path)."""

    def __init__(self, world_dims, dim_in, dim_out, emb_dim, dropout=0.1, use_text_condition=False):
        super().__init__()
        assert world_dims == 3 and not use_text_condition
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, dim_out))
        self.block1 = nn.Sequential(convnormalization(dim_in), nn.SiLU(),
                                    nn.Conv3d(dim_in, dim_out, 3, padding=1))
        conv2 = nn.Conv3d(dim_out, dim_out, 3, padding=1)
        for p in conv2.parameters():
            p.detach().zero_()
        self.block2 = nn.Sequential(convnormalization(dim_out), nn.SiLU(), nn.Dropout(dropout), conv2)
        self.res_conv = nn.Conv3d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()

    def forward(self, x, time_emb, text_condition=None):
        h = self.block1(x)
        h = h + self.time_mlp(time_emb)[:, :, None, None, None]
        h = self.block2(h)
        return h + self.res_conv(x)


class QKVAttention(nn.Module):
    """reference modules.py:538-547."""

    def forward(self, qkv):
        ch = qkv.shape[1] // 3
        q, k, v = torch.split(qkv, ch, dim=1)
        scale = 1 / math.sqrt(math.sqrt(ch))
        weight = torch.einsum('bct,bcs->bts', q * scale, k * scale)
        weight = torch.softmax(weight.float(), dim=-1).type(weight.dtype)
        return torch.einsum('bts,bcs->bct', weight, v)


class AttentionBlock(nn.Module):
    """reference modules.py:515-535."""

    def __init__(self, channels, num_heads=1):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads
        self.norm = convnormalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = QKVAttention()
        self.proj_out = nn.Conv1d(channels, channels, 1)
        for p in self.proj_out.parameters():
            p.detach().zero_()

    def forward(self, x):
        b, c, *spatial = x.shape
        x = x.reshape(b, c, -1)
        qkv = self.qkv(self.norm(x))
        qkv = qkv.reshape(b * self.num_heads, -1, qkv.shape[2])
        h = self.attention(qkv).reshape(b, -1, qkv.shape[2])
        return (x + self.proj_out(h)).reshape(b, c, *spatial)


class LearnedSinusoidalPosEmb(nn.Module):
    """reference modules.py:550-563."""

    def __init__(self, dim):
        super().__init__()
        assert dim % 2 == 0
        self.weights = nn.Parameter(torch.randn(dim // 2))

    def forward(self, x):
        x = x[:, None]
        freqs = x * self.weights[None, :] * 2 * math.pi
        return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


class UNet3DModel(nn.Module):
    def __init__(self, full_depth, in_split_channels, model_channels, out_split_channels,
                 attention_resolutions, dropout=0, channel_mult=(1, 2, 4, 8), dims=2, num_classes=None,
                 use_checkpoint=False, num_heads=-1, use_text_condition=False, context_dim=None,
                 n_embed=None, **kwargs):
        super().__init__()
        assert dims == 3
        self.full_depth = full_depth
        self.in_channels = in_split_channels
        self.model_channels = model_channels
        self.out_channels = out_split_channels
        self.attention_resolutions = attention_resolutions
        self.channel_mult = list(channel_mult)
        self.num_classes = num_classes
        self.num_heads = num_heads
        mc = model_channels
        chans = [mc] + [mc * m for m in self.channel_mult]
        in_out = list(zip(chans[:-1], chans[1:]))
        ted = mc * 4
        self.time_pos_emb = LearnedSinusoidalPosEmb(mc)
        self.time_emb = nn.Sequential(nn.Linear(mc + 1, ted), nn.SiLU(), nn.Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)
        self.input_emb = nn.Conv3d(2 * self.in_channels, mc, 3, padding=1)

        def attn(c, ds):
            if ds in attention_resolutions:
                return nn.Sequential(convnormalization(c), nn.SiLU(), AttentionBlock(c, num_heads=num_heads))
            return our_Identity()

        self.downs = nn.ModuleList()
        self.ups = nn.ModuleList()
        nres = len(in_out)
        ds = 1
        for ind, (ci, co) in enumerate(in_out):
            last = ind >= nres - 1
            self.downs.append(nn.ModuleList([
                ResnetBlock(dims, ci, co, emb_dim=ted, dropout=dropout), attn(co, ds),
                ConvDownsample(co, dims=dims) if not last else our_Identity()]))
            if not last:
                ds *= 2
        mid = chans[-1]
        self.mid_block1 = ResnetBlock(dims, mid, mid, emb_dim=ted, dropout=dropout)
        self.mid_self_attn = attn(mid, ds)
        self.mid_block2 = ResnetBlock(dims, mid, mid, emb_dim=ted, dropout=dropout)
        # graph_unet_lr.py:152-166: walks reversed(in_out[1:]); every level upsamples
        for ci, co in reversed(in_out[1:]):
            self.ups.append(nn.ModuleList([
                ResnetBlock(dims, co * 2, ci, emb_dim=ted, dropout=dropout), attn(ci, ds),
                ConvUpsample(ci, dims=dims)]))
            ds //= 2
        self.end = nn.Sequential(convnormalization(mc), nn.SiLU())
        self.out = nn.Conv3d(mc, self.out_channels, 3, padding=1)

    @torch.no_grad()
    def forward_as_middle(self, h, doctree, timesteps, label, context):
        vox = ops.octree2voxel_cf(h, doctree.batch_size, self.full_depth)       # [B, C, S, S, S]
        vox = self.forward(x=vox, timesteps=timesteps, label=label, context=context, as_middle=True)
        return ops.voxel2octree_cf(vox, self.full_depth)

    @torch.no_grad()
    def forward(self, x=None, timesteps=None, x_self_cond=None, label=None, context=None, as_middle=False,
                **kwargs):
        assert (label is not None) == (self.num_classes is not None), \
            'must specify label if and only if the model is class-conditional'
        if not as_middle:
            if x_self_cond is None:
                x_self_cond = torch.zeros_like(x)
            x = self.input_emb(torch.cat((x, x_self_cond), dim=1))
        emb = self.time_emb(self.time_pos_emb(timesteps))
        if self.num_classes is not None:
            assert label.shape == (x.shape[0],)
            emb = emb + self.label_emb(label)
        hs = []
        for resnet, self_attn, downsample in self.downs:
            x = self_attn(resnet(x, emb))
            hs.append(x)
            x = downsample(x)
        x = self.mid_block2(self.mid_self_attn(self.mid_block1(x, emb)), emb)
        for resnet, self_attn, upsample in self.ups:
            x = upsample(self_attn(resnet(torch.cat((x, hs.pop()), dim=1), emb)))
        x = self.end(x)
        return x if as_middle else self.out(x)
