"""Sparse (dual-octree) diffusion U-Net -- stage "hr" / "feature".

Mirror of reference models/networks/diffusion_networks/graph_unet_hr.py
(``UNet3DModel``: ctor :69-209, forward :214-281): same constructor keywords,
same ``forward`` / ``forward_as_middle`` signatures, same state_dict keys
(time_embed.*, label_emb.*, input_blocks.N.*, middle_block1/2.*,
output_blocks.N.*, end_norm.*, out.weights).  The block list is generated from a
level plan instead of the reference's nested loops; every block runs on libofx.
"""
import torch
import torch.nn as nn

from . import ops
from .modules import (GraphConv, GraphDownsample, GraphResBlockEmbed, GraphUpsample, _Linear,
                      graphnormalization)


def level_plan(input_depth, channel_mult, num_res_blocks):
    """[(kind, depth_of_input, mult_index)] for the encoder and decoder halves."""
    enc, dec = [], []
    d = input_depth
    last = len(channel_mult) - 1
    for level in range(len(channel_mult)):
        enc += [('res', d, level)] * num_res_blocks[level]
        if level != last:
            enc.append(('down', d, level))
            d -= 1
    for level in range(last, -1, -1):
        for i in range(num_res_blocks[level] + 1):
            dec.append(('res', d, level))
            if level and i == num_res_blocks[level]:
                dec.append(('up', d, level))
                d += 1
    return enc, dec


class UNet3DModel(nn.Module):
    def __init__(self, image_size, input_depth, full_depth, in_channels, model_channels, lr_model_channels,
                 out_channels, num_res_blocks, dropout=0, channel_mult=(1, 2, 4), dims=3, num_classes=None,
                 use_checkpoint=False, num_heads=-1, use_scale_shift_norm=False, **kwargs):
        super().__init__()
        self.image_size = image_size
        self.input_depth = input_depth
        self.full_depth = full_depth
        self.in_channels = in_channels
        self.model_channels = model_channels
        self.out_channels = out_channels
        self.num_res_blocks = list(num_res_blocks)
        self.dropout = dropout
        self.channel_mult = list(channel_mult)
        self.num_classes = num_classes
        self.use_checkpoint = use_checkpoint
        self.dtype = torch.float32
        self.num_heads = num_heads
        et, deg = 7, 7
        ted = model_channels * 4

        self.time_embed = nn.Sequential(_Linear(model_channels, ted), nn.SiLU(), _Linear(ted, ted))
        if num_classes is not None:
            self.label_emb = nn.Embedding(num_classes, ted)

        enc, dec = level_plan(input_depth, self.channel_mult, self.num_res_blocks)
        self._enc, self._dec = enc, dec

        def res(cin, cout, d):
            return GraphResBlockEmbed(cin, ted, dropout, cout, et, deg, d - 1, dims=dims,
                                      use_checkpoint=use_checkpoint,
                                      use_scale_shift_norm=use_scale_shift_norm)

        blocks = [GraphConv(in_channels, model_channels, et, deg, input_depth - 1)]
        skip_ch = [model_channels]
        ch = model_channels
        d = input_depth
        for kind, d, level in enc:
            if kind == 'res':
                co = self.channel_mult[level] * model_channels
                blocks.append(res(ch, co, d))
                ch = co
            else:   # GraphDownsample(d -> d-1): its conv lives at depth d-1, node types (d-1)-1
                blocks.append(GraphDownsample(ch, ch, et, deg, d - 2))
            skip_ch.append(ch)
        self.input_blocks = nn.ModuleList(blocks)
        d_mid = input_depth - (len(self.channel_mult) - 1)
        self._d_mid = d_mid
        self.middle_block1 = res(ch, lr_model_channels, d_mid)
        self.middle_block2 = res(lr_model_channels * 2, ch, d_mid)

        outs = []
        for kind, d, level in dec:
            if kind == 'res':
                co = model_channels * self.channel_mult[level]
                outs.append(res(ch + skip_ch.pop(), co, d))
                ch = co
            else:   # GraphUpsample(d -> d+1): conv at depth d+1, node types (d+1)-1
                outs.append(GraphUpsample(ch, ch, et, deg, d))
        self.output_blocks = nn.ModuleList(outs)

        self.end_norm = graphnormalization(ch)
        self.end = nn.SiLU()
        self.out = GraphConv(ch, out_channels, et, deg, input_depth - 1)
        self.out.emit_stats = False
        for p in self.out.parameters():          # zero_module (graph_unet_hr.py:209)
            p.detach().zero_()

    def forward_as_middle(self, h, doctree, timesteps, label, context):
        return self.forward(x=h, doctree=doctree, timesteps=timesteps, label=label, context=context,
                            as_middle=True)

    @torch.no_grad()
    def forward(self, x=None, doctree=None, unet_lr=None, timesteps=None, label=None, context=None,
                as_middle=False, **kwargs):
        assert (label is not None) == (self.num_classes is not None), \
            'must specify y if and only if the model is class-conditional'
        t_emb = ops.timestep_embedding(timesteps.float(), self.model_channels)
        emb = self.time_embed[0](t_emb)
        emb = self.time_embed[2](ops.act(emb, 'silu'))
        if self.num_classes is not None:
            assert label.shape == (doctree.batch_size,)
            emb = emb + self.label_emb(label)
        emb_act = ops.act(emb, 'silu')          # SiLU(emb) is what every res-block consumes

        d = self.input_depth
        h = x if as_middle else self.input_blocks[0](x, doctree, d)
        hs = [h]
        for (kind, dd, _), module in zip(self._enc, self.input_blocks[1:]):
            if kind == 'res':
                h = module(h, emb, doctree, dd, emb_act=emb_act)
            else:
                h = module(h, doctree, dd)
            hs.append(h)
        d = self._d_mid

        if unet_lr is not None:
            h = self.middle_block1(h, emb, doctree, d, emb_act=emb_act)
            h_lr = unet_lr.forward_as_middle(h, doctree, timesteps, label, context)
            h = ops.cat_channels(h, h_lr)
            h = self.middle_block2(h, emb, doctree, d, emb_act=emb_act)

        for (kind, dd, _), module in zip(self._dec, self.output_blocks):
            if kind == 'res':
                h = ops.cat_channels(h, hs.pop())
                h = module(h, emb, doctree, dd, emb_act=emb_act)
            else:
                h = module(h, doctree, dd)
        h = self.end_norm(h, doctree, self.input_depth, act='silu')
        if as_middle:
            return h
        out = self.out(h, doctree, self.input_depth)
        assert out.shape[0] == x.shape[0]
        return out
