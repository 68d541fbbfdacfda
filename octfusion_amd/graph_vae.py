"""GraphVAE -- decoder side (the part of the sampling path that turns the generated latent
code into per-node SDF coefficients and grows the octree to its final depth).

Mirror of reference models/networks/dualoctree_networks/graph_vae.py (ctor :52-124,
octree_decoder :171-223, create_child_octree :236-244, decode_code :300-324) and the VAE
flavoured GraphDownsample / GraphUpsample (dualoctree_networks/modules.py:39-95).  The full
module tree (encoder included) is constructed so a reference VAE checkpoint loads with
strict=True; only the decoder has a forward here (the encoder runs at training time only,
SURVEY.md section 8: out of scope).  NeuralMPU SDF evaluation is the next row (section 8f).
"""
import torch
import torch.nn as nn

from . import mpu, ops
from .dual_octree import DualOctree
from .modules import (Conv1x1, Conv1x1GnGelu, Conv1x1GnGeluSequential, Downsample, DualOctreeGroupNorm,
                      GraphConv, GraphResBlocks, Upsample, pool_nodes, unpool_nodes)
from .octree import Octree

CHANNELS = [4, 512, 512, 256, 128, 64, 32, 32, 24, 8]      # graph_vae.py:125


class GraphDownsample(nn.Module):
    """VAE flavour (dualoctree_networks/modules.py:39-68): forward(x, doctree, d, leaf_mask, numd, lnumd),
    `d` = depth of the OUTPUT; the masks / counts are accepted for call compatibility and ignored
    (the doctree's cached row maps replace them)."""

    def __init__(self, channels_in, channels_out=None):
        super().__init__()
        self.channels_in = channels_in
        self.channels_out = channels_out or channels_in
        self.downsample = Downsample(channels_in)
        if self.channels_in != self.channels_out:
            self.conv1x1 = Conv1x1GnGelu(self.channels_in, self.channels_out)

    def forward(self, x, doctree, d, leaf_mask=None, numd=None, lnumd=None):
        out = pool_nodes(x, doctree, d + 1, self.downsample)
        if self.channels_in != self.channels_out:
            out = self.conv1x1(out, doctree, d)
        return out


class GraphUpsample(nn.Module):
    """VAE flavour (dualoctree_networks/modules.py:71-95): forward(x, doctree, d, leaf_mask, numd)."""

    def __init__(self, channels_in, channels_out=None):
        super().__init__()
        self.channels_in = channels_in
        self.channels_out = channels_out or channels_in
        self.upsample = Upsample(channels_in)
        if self.channels_in != self.channels_out:
            self.conv1x1 = Conv1x1GnGelu(self.channels_in, self.channels_out)

    def forward(self, x, doctree, d, leaf_mask=None, numd=None):
        out = unpool_nodes(x, doctree, d - 1, self.upsample)
        if self.channels_in != self.channels_out:
            out = self.conv1x1(out, doctree, d)
        return out


class GraphVAE(nn.Module):
    def __init__(self, depth, channel_in, nout, full_depth=2, depth_stop=6, depth_out=8, use_checkpoint=False,
                 resblk_type='bottleneck', bottleneck=4, resblk_num=3, code_channel=3, embed_dim=3):
        super().__init__()
        self.depth, self.channel_in, self.nout = depth, channel_in, nout
        self.full_depth, self.depth_stop, self.depth_out = full_depth, depth_stop, depth_out
        self.neural_mpu = mpu.NeuralMPU(full_depth, depth_stop, depth_out)          # graph_vae.py:69
        self.resblk_num = resblk_num
        self.channels = CHANNELS
        self.resblk_nums = [resblk_num] * 16
        ch, rn = self.channels, self.resblk_nums
        et, deg = 7, 7
        # encoder (parameters only: checkpoint compatibility)
        self.conv1 = GraphConv(channel_in, ch[depth], et, deg, depth - 1)
        self.encoder = nn.ModuleList([GraphResBlocks(ch[d], ch[d], 0.0, rn[d] - 1, et, deg, d - 1, use_checkpoint)
                                      for d in range(depth, depth_stop - 1, -1)])
        self.downsample = nn.ModuleList([GraphDownsample(ch[d], ch[d - 1]) for d in range(depth, depth_stop, -1)])
        self.encoder_norm_out = DualOctreeGroupNorm(ch[depth_stop])
        self.nonlinearity = nn.GELU()
        # decoder
        self.decoder = nn.ModuleList([GraphResBlocks(ch[d], ch[d], 0.0, rn[d], et, deg, d - 1, use_checkpoint)
                                      for d in range(depth_stop, depth + 1)])
        self.decoder_mid = nn.Module()
        self.decoder_mid.block_1 = GraphResBlocks(ch[depth_stop], ch[depth_stop], 0.0, rn[depth_stop], et, deg,
                                                  depth_stop - 1, use_checkpoint)
        self.decoder_mid.block_2 = GraphResBlocks(ch[depth_stop], ch[depth_stop], 0.0, rn[depth_stop], et, deg,
                                                  depth_stop - 1, use_checkpoint)
        self.upsample = nn.ModuleList([GraphUpsample(ch[d - 1], ch[d]) for d in range(depth_stop + 1, depth + 1)])
        self.predict = nn.ModuleList([self._make_predict_module(ch[d], 2) for d in range(depth_stop, depth + 1)])
        self.regress = nn.ModuleList([self._make_predict_module(ch[d], 4) for d in range(depth_stop, depth + 1)])
        self.code_channel = code_channel
        self.KL_conv = Conv1x1(ch[depth_stop], 2 * embed_dim, use_bias=True)
        self.post_KL_conv = Conv1x1(embed_dim, ch[depth_stop], use_bias=True)

    def _make_predict_module(self, channel_in, channel_out=2, num_hidden=32):
        return nn.Sequential(Conv1x1GnGeluSequential(channel_in, num_hidden),
                             Conv1x1(num_hidden, channel_out, use_bias=True))

    def create_full_octree(self, octree_in):
        octree = Octree(self.depth, self.full_depth, octree_in.batch_size, octree_in.device)
        for d in range(self.full_depth + 1):
            octree.octree_grow_full(d)
        return octree

    def create_child_octree(self, octree_in):
        octree_out = self.create_full_octree(octree_in)
        octree_out.depth = self.full_depth
        for d in range(self.full_depth, self.depth_stop):
            octree_out.octree_split(octree_in.nempty_mask(d).to(torch.int32), d)
            octree_out.octree_grow(d + 1)
            octree_out.depth += 1
        return octree_out

    @torch.no_grad()
    def octree_encoder_step(self, data, doctree):
        """graph_vae.py:134-160 with the input feature passed in (the reference takes it from the ocnn octree,
        graph_vae.py:131-132): data [N_depth, channel_in] -> {d: features} down to depth_stop."""
        depth, ds = self.depth, self.depth_stop
        convs = {depth: data}
        for i, d in enumerate(range(depth, ds - 1, -1)):
            convd = convs[d]
            if d == depth:
                convd = self.conv1(convd, doctree, d)
            convd = self.encoder[i](convd, doctree, d)
            convs[d] = convd
            if d > ds:
                convs[d - 1] = self.downsample[i](convd, doctree, d - 1)
        convs[ds] = self.encoder_norm_out(convs[ds], doctree, ds, act='gelu')
        return convs

    @torch.no_grad()
    def encode(self, data, doctree, noise=None, sample=True):
        """graph_vae.py:162-170 / 291-298: KL_conv -> DiagonalGaussianDistribution (distributions.py:24-37).
        Returns (code [N, embed_dim], mean, logvar); code = mean + exp(logvar / 2) * noise, or the mean."""
        h = self.octree_encoder_step(data, doctree)[self.depth_stop]
        mean, logvar = torch.chunk(self.KL_conv(h), 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        if not sample:
            return mean.contiguous(), mean, logvar
        if noise is None:
            noise = torch.randn_like(mean)
        return (mean + torch.exp(0.5 * logvar) * noise).contiguous(), mean, logvar

    @torch.no_grad()
    def octree_decoder(self, code, doctree_out, update_octree=False):
        ds = self.depth_stop
        x = self.post_KL_conv(code)
        x = self.decoder_mid.block_1(x, doctree_out, ds)
        x = self.decoder_mid.block_2(x, doctree_out, ds)
        logits, reg_voxs = {}, {}
        deconv = x
        for i, d in enumerate(range(ds, self.depth_out + 1)):
            if d > ds:
                deconv = self.upsample[i - 1](deconv, doctree_out, d)
            deconv = self.decoder[i](deconv, doctree_out, d)
            logit = self.predict[i][1](self.predict[i][0]((deconv, doctree_out, d)))
            nnum = int(doctree_out.nnum[d])
            logits[d] = logit[logit.shape[0] - nnum:]
            if update_octree:
                label = logits[d].argmax(1).to(torch.int32)
                octree_out = doctree_out.octree
                octree_out.octree_split(label, d)
                if d < self.depth_out:
                    octree_out.octree_grow(d + 1)
                    octree_out.depth += 1
                doctree_out = DualOctree(octree_out, prev=doctree_out)       # only the new depth is built
            reg = self.regress[i][1](self.regress[i][0]((deconv, doctree_out, d)))
            # pad to [leaves-so-far(all nodes) + nodes at d] rows (graph_vae.py:214-221) with a row map
            node_mask = doctree_out.graph[d]['node_mask']
            pad = torch.zeros(node_mask.shape[0], reg.shape[1], dtype=torch.float32, device=reg.device)
            dmap = doctree_out.pad_rows(d)
            ops.rows_copy(reg, pad, reg.shape[0], dmap=dmap)
            reg_voxs[d] = pad
        return logits, reg_voxs, doctree_out.octree

    @torch.no_grad()
    def decode_code(self, code, doctree_in, update_octree=True, pos=None):
        if update_octree:
            octree_out = self.create_child_octree(doctree_in.octree)
            doctree_out = DualOctree(octree_out)
        else:
            doctree_out = doctree_in
        out = self.octree_decoder(code, doctree_out, update_octree=update_octree)
        output = {'logits': out[0], 'reg_voxs': out[1], 'octree_out': out[2]}
        if pos is not None:                                     # graph_vae.py:316-317
            output['mpus'] = self.neural_mpu(pos, out[1], out[2])
        # graph_vae.py:319-323: the SDF field of the decoded shape (callable on pts [n,4]; calc_sdf sweeps it)
        output['neural_mpu'] = mpu.MpuField(self.full_depth, self.depth_out, out[1][self.depth_out], out[2])
        return output
