"""Sparse (dual-octree) diffusion U-Net -- stage "hr" / "feature".

Mirror of reference models/networks/diffusion_networks/graph_unet_hr.py
(``UNet3DModel``: ctor :69-209, forward :214-281): same constructor keywords,
same ``forward`` / ``forward_as_middle`` signatures, same state_dict keys
(time_embed.*, label_emb.*, input_blocks.N.*, middle_block1/2.*,
output_blocks.N.*, end_norm.*, out.weights).  The block list is generated from a
level plan instead of the reference's nested loops; every block runs on libofx.
"""
import inspect

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
        self._skip_ch_all = list(skip_ch)
        self._mid_out_ch = ch
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

        # (C_h, C_skip, depth) of the decoder concat that consumes skip i (production order), for forward()
        skip_depths = [input_depth] + [(dd if kind == 'res' else dd - 1) for kind, dd, _ in enc]
        plan = []
        ch_run = self._mid_out_ch
        sk = list(self._skip_ch_all)
        for kind, dd, level in dec:
            if kind == 'res':
                c_skip = sk.pop()
                plan.append((ch_run, c_skip, dd))
                ch_run = model_channels * self.channel_mult[level]
        self._cat_plan = list(reversed(plan))
        assert [p_[2] for p_ in self._cat_plan] == skip_depths, (self._cat_plan, skip_depths)
        self.end_norm = graphnormalization(ch)
        self.end = nn.SiLU()
        self.out = GraphConv(ch, out_channels, et, deg, input_depth - 1)
        self.out.emit_stats = False
        for p in self.out.parameters():          # zero_module (graph_unet_hr.py:209)
            p.detach().zero_()

    def _all_emb_outs(self, emb_act):
        """{id(block): emb_layers(emb)[B, Cout]} for every GraphResBlockEmbed of this net, from ONE GEMM."""
        blocks = [m for m in self.modules() if isinstance(m, GraphResBlockEmbed)]
        lins = [b.emb_layers[1] for b in blocks]
        key = tuple((l.weight.data_ptr(), l.weight._version, l.bias.data_ptr(), l.bias._version) for l in lins)
        if getattr(self, '_emb_cat_key', None) != key:
            self._emb_cat_w = torch.cat([l.weight.detach() for l in lins], dim=0).contiguous()
            self._emb_cat_b = torch.cat([l.bias.detach() for l in lins], dim=0).contiguous()
            self._emb_cat_pw = ops.PackedWeight()
            self._emb_cat_key = key
        if ops.LINEAR_SMALL and emb_act.shape[0] <= 16:
            allout = ops.linear_small(emb_act, self._emb_cat_w, self._emb_cat_b)
        else:
            allout = ops.gemm(emb_act, self._emb_cat_pw.get(self._emb_cat_w, 'nk'), self._emb_cat_b)
        outs, off = {}, 0
        for b, l in zip(blocks, lins):
            outs[id(b)] = allout[:, off:off + l.out_features]
            off += l.out_features
        return outs

    def forward_as_middle(self, h, doctree, timesteps, label, context, out=None):
        """reference graph_unet_hr.py:211-212; ``out``: optional destination rows (a column slice of the caller's
        concatenation buffer)."""
        return self.forward(x=h, doctree=doctree, timesteps=timesteps, label=label, context=context,
                            as_middle=True, out=out)

    @torch.no_grad()
    def forward(self, x=None, doctree=None, unet_lr=None, timesteps=None, label=None, context=None,
                as_middle=False, out=None, **kwargs):
        with ops.stats_scope(x.device):           # one zero-fill for every fused GroupNorm statistics buffer
            return self._forward(x, doctree, unet_lr, timesteps, label, context, as_middle, out)

    def _forward(self, x, doctree, unet_lr, timesteps, label, context, as_middle, out_mid=None):
        assert (label is not None) == (self.num_classes is not None), \
            'must specify y if and only if the model is class-conditional'
        timesteps = timesteps.float()
        emb = None                              # (the blocks get SiLU(emb) and their own projection of it)

        def embed():
            t_emb = ops.timestep_embedding(timesteps, self.model_channels)
            # time_embed = Linear -> SiLU -> Linear (+ label embedding), then every res-block applies SiLU to it
            # (modules.py:754): the activations ride in the few-row linear launches
            lab = None
            if self.num_classes is not None:
                assert label.shape == (doctree.batch_size,)
                lab = self.label_emb(label)
            e_act = self.time_embed[2](self.time_embed[0](t_emb, act_out='silu'), res=lab, act_out='silu')
            # every res-block applies its own Linear(ted -> Cout) to the same SiLU(emb) (modules.py:754): one GEMM
            # against the row-concatenated weights instead of one launch per block
            return e_act, self._all_emb_outs(e_act)
        # the embedding chains of this net and of the nested one depend on the timesteps only: parallel branches beside
        # the input convolution / the encoder (ops.fork_stream), joined where the first res-block needs them
        emb_main, emb_fork = ops.fork_stream(x.device)
        if emb_fork is not None:
            with torch.cuda.stream(emb_fork):
                emb_act, emb_outs = embed()
            if unet_lr is not None and hasattr(unet_lr, 'precompute_embeddings'):
                unet_lr.precompute_embeddings(timesteps, label, doctree.batch_size)
        else:
            emb_act, emb_outs = embed()

        # Zero-copy skip concatenation: the decoder block that consumes skip tensor i reads ONE buffer
        # [N, C_h + C_skip]; the encoder module that produces the skip writes straight into its right
        # columns and the decoder module that produces h into its left columns (every libofx kernel takes
        # a leading dimension), so torch.cat never copies activations.
        dev = x.device
        n_at = {dd: doctree.csr(dd)[2] for dd in range(self._d_mid, self.input_depth + 1)}
        cat_bufs = []                     # one per skip tensor, in production order
        for (c_h, c_skip, dd) in self._cat_plan:
            cat_bufs.append(torch.empty(n_at[dd], c_h + c_skip, dtype=torch.float32, device=dev))

        def skip_slot(i):
            c_h, c_skip, _ = self._cat_plan[i]
            return cat_bufs[i][:, c_h:]

        d = self.input_depth
        if as_middle:
            h = x
            ops.rows_copy(x, skip_slot(0), x.shape[0])
            hs = [skip_slot(0)]
        else:
            h = self.input_blocks[0](x, doctree, d, out=skip_slot(0))
            hs = [h]
        if emb_fork is not None:
            emb_main.wait_stream(emb_fork)
        for k, ((kind, dd, _), module) in enumerate(zip(self._enc, self.input_blocks[1:])):
            slot = skip_slot(k + 1)
            if kind == 'res':
                h = module(h, emb, doctree, dd, emb_act=emb_act, out=slot, emb_out=emb_outs[id(module)])
            else:
                h = module(h, doctree, dd, out=slot)
            hs.append(h)
        d = self._d_mid

        if unet_lr is not None:
            # cat([h, h_lr], dim=1) (graph_unet_hr.py:252) without the copy: middle_block1 writes the left columns of one
            # buffer, the nested net's last GroupNorm the right ones (`out=` is this package's extension of
            # forward_as_middle; a nested net without it gets a plain concatenation)
            c_mid = self.middle_block1.out_channels
            mid = torch.empty(n_at[d], 2 * c_mid, dtype=torch.float32, device=dev)
            h = self.middle_block1(h, emb, doctree, d, emb_act=emb_act, out=mid[:, :c_mid], emb_out=emb_outs[id(self.middle_block1)])
            if 'out' in inspect.signature(unet_lr.forward_as_middle).parameters:
                h_lr = unet_lr.forward_as_middle(h, doctree, timesteps, label, context, out=mid[:, c_mid:])
            else:
                h_lr = unet_lr.forward_as_middle(h, doctree, timesteps, label, context)
            if h_lr.data_ptr() == mid[:, c_mid:].data_ptr() and h_lr.shape[1] == c_mid:
                h = ops.cat_channels(h, h_lr, buf=mid)
            else:
                h = ops.cat_channels(h, h_lr)
            h = self.middle_block2(h, emb, doctree, d, emb_act=emb_act, emb_out=emb_outs[id(self.middle_block2)])

        # decoder: block j consumes skip len(hs)-1-j; its own output goes to the left columns of the NEXT
        # consumer's buffer when the next module is a res block
        n_skip = len(hs)
        j = 0                             # skips consumed so far
        pending = h                       # tensor that will become the left half of the next concat
        for idx, ((kind, dd, _), module) in enumerate(zip(self._dec, self.output_blocks)):
            nxt_is_res = idx + 1 < len(self._dec) and self._dec[idx + 1][0] == 'res'
            out_slot = None
            if nxt_is_res:
                c_h_n = self._cat_plan[n_skip - 1 - (j + (1 if kind == 'res' else 0))][0]
                out_slot = cat_bufs[n_skip - 1 - (j + (1 if kind == 'res' else 0))][:, :c_h_n]
            if kind == 'res':
                si = n_skip - 1 - j
                c_h = self._cat_plan[si][0]
                left = cat_bufs[si][:, :c_h]
                if pending.data_ptr() != left.data_ptr():          # first decoder block: h came from the middle
                    ops.rows_copy(pending, left, pending.shape[0])
                    st = ops.get_stats(pending)
                    if st is not None:
                        setattr(left, ops.STATS_ATTR, st)
                    pending = left
                hcat = ops.cat_channels(pending, hs[si], buf=cat_bufs[si])
                pending = module(hcat, emb, doctree, dd, emb_act=emb_act, out=out_slot, emb_out=emb_outs[id(module)])
                j += 1
            else:
                pending = module(pending, doctree, dd, out=out_slot)
        h = pending
        h = self.end_norm(h, doctree, self.input_depth, act='silu', out=out_mid if as_middle else None)
        if as_middle:
            return h
        out = self.out(h, doctree, self.input_depth)
        assert out.shape[0] == x.shape[0]
        return out
