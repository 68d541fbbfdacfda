"""Oracle: GraphVAE decoder (torch CPU).  TEST INFRASTRUCTURE.

Restates reference models/networks/dualoctree_networks/graph_vae.py:171-223
(octree_decoder), :236-244 (create_child_octree), :300-324 (decode_code) and the VAE
flavoured GraphUpsample (dualoctree_networks/modules.py:71-95) over a reference
state_dict.  Channel table graph_vae.py:125.
"""
import torch
import torch.nn.functional as F

from . import modules as M
from .dual_octree import OracleDualOctree
from .octree import Octree

CHANNELS = [4, 512, 512, 256, 128, 64, 32, 32, 24, 8]      # graph_vae.py:125


def _head(x, doctree, d, sd):
    """_make_predict_module (graph_vae.py:127-130): Conv1x1GnGeluSequential -> Conv1x1(bias)."""
    h = M.conv1x1_gn(x, doctree, d, M._sub(sd, '0'), gelu=True)
    return F.linear(h, sd['1.linear.weight'], sd['1.linear.bias'])


def vae_upsample(x, doctree, d, sd):
    """dualoctree_networks/modules.py:82-90; `d` = depth of the OUTPUT (the reference passes it so)."""
    out = M.unpool_rearrange(x, doctree, d - 1, sd['upsample.weights'])
    if 'conv1x1.conv.linear.weight' in sd:
        out = M.conv1x1_gn(out, doctree, d, M._sub(sd, 'conv1x1'), gelu=True)
    return out


def create_child_octree(octree_in, depth, full_depth, depth_stop):
    """graph_vae.py:225-244."""
    out = Octree(depth, full_depth, octree_in.batch_size)
    for d in range(full_depth + 1):
        out.octree_grow_full(d)
    out.depth = full_depth
    for d in range(full_depth, depth_stop):
        out.octree_split(octree_in.nempty_mask(d).long(), d)
        out.octree_grow(d + 1)
        out.depth += 1
    return out


def octree_decoder(sd, cfg, code, doctree_out, update_octree=False):
    """graph_vae.py:171-223."""
    depth_stop, depth_out = cfg['depth_stop'], cfg['depth_out']
    x = F.linear(code, sd['post_KL_conv.linear.weight'], sd['post_KL_conv.linear.bias'])
    x = M.graph_resblocks(x, doctree_out, depth_stop, M._sub(sd, 'decoder_mid.block_1'), depth_stop - 1)
    x = M.graph_resblocks(x, doctree_out, depth_stop, M._sub(sd, 'decoder_mid.block_2'), depth_stop - 1)
    logits, reg_voxs = {}, {}
    deconv = x
    for i, d in enumerate(range(depth_stop, depth_out + 1)):
        if d > depth_stop:
            deconv = vae_upsample(deconv, doctree_out, d, M._sub(sd, 'upsample.%d' % (i - 1)))
        deconv = M.graph_resblocks(deconv, doctree_out, d, M._sub(sd, 'decoder.%d' % i), d - 1)
        logit = _head(deconv, doctree_out, d, M._sub(sd, 'predict.%d' % i))
        nnum = int(doctree_out.nnum[d])
        logits[d] = logit[-nnum:]
        if update_octree:
            label = logits[d].argmax(1).to(torch.int32)
            octree_out = doctree_out.octree
            octree_out.octree_split(label, d)
            if d < depth_out:
                octree_out.octree_grow(d + 1)
                octree_out.depth += 1
            doctree_out = OracleDualOctree(octree_out)
            doctree_out.post_processing_for_docnn()
        reg = _head(deconv, doctree_out, d, M._sub(sd, 'regress.%d' % i))
        node_mask = doctree_out.graph[d]['node_mask']
        pad = torch.zeros(node_mask.shape[0], reg.shape[1])
        pad[node_mask] = reg
        reg_voxs[d] = pad
    return logits, reg_voxs, doctree_out.octree


def decode_code(sd, cfg, code, doctree_in, update_octree=True):
    """graph_vae.py:300-324 (without the MPU closure)."""
    if update_octree:
        octree_out = create_child_octree(doctree_in.octree, cfg['depth'], cfg['full_depth'], cfg['depth_stop'])
        doctree_out = OracleDualOctree(octree_out)
        doctree_out.post_processing_for_docnn()
    else:
        doctree_out = doctree_in
    return octree_decoder(sd, cfg, code, doctree_out, update_octree)


def vae_downsample(x, doctree, d, sd):
    """dualoctree_networks/modules.py:39-68; `d` = depth of the OUTPUT."""
    out = M.pool_rearrange(x, doctree, d + 1, sd['downsample.weights'])
    if 'conv1x1.conv.linear.weight' in sd:
        out = M.conv1x1_gn(out, doctree, d, M._sub(sd, 'conv1x1'), gelu=True)
    return out


def octree_encoder_step(sd, cfg, data, doctree):
    """graph_vae.py:134-160 with the input feature given; returns {d: features}."""
    depth, ds = cfg['depth'], cfg['depth_stop']
    convs = {depth: data}
    for i, d in enumerate(range(depth, ds - 1, -1)):
        convd = convs[d]
        if d == depth:
            convd = M.graph_conv(convd, doctree, d, sd['conv1.weights'], None, depth - 1)
        convd = M.graph_resblocks(convd, doctree, d, M._sub(sd, 'encoder.%d' % i), d - 1)
        convs[d] = convd
        if d > ds:
            convs[d - 1] = vae_downsample(convd, doctree, d - 1, M._sub(sd, 'downsample.%d' % i))
    h = M.dual_octree_group_norm(convs[ds], doctree, ds, sd['encoder_norm_out.weights'], sd['encoder_norm_out.bias'])
    convs[ds] = F.gelu(h)
    return convs


def encode(sd, cfg, data, doctree):
    """graph_vae.py:162-170: the KL_conv output [N, 2 * embed_dim] (mean | logvar)."""
    h = octree_encoder_step(sd, cfg, data, doctree)[cfg['depth_stop']]
    return h, F.linear(h, sd['KL_conv.linear.weight'], sd['KL_conv.linear.bias'])


def forward_train(sd, cfg, data, doctree_in, doctree_out, pos, noise):
    """GraphVAE.forward with a ground-truth output octree (graph_vae.py:246-289, update_octree False):
    encoder -> posterior sample with the given noise -> decoder -> NeuralMPU at `pos`.
    Returns {'logits', 'reg_voxs', 'mpus', 'kl' (elementwise, distributions.py:46), 'z'}."""
    from . import loss as OL
    from . import mpu as OM
    _, params = encode(sd, cfg, data, doctree_in)
    z, kl = OL.posterior(params, noise)
    logits, reg_voxs, octree_out = octree_decoder(sd, cfg, z, doctree_out, update_octree=False)
    mpus = OM.neural_mpu(pos, reg_voxs, octree_out, cfg['full_depth'], cfg['depth_stop'], cfg['depth_out'])
    return {'logits': logits, 'reg_voxs': reg_voxs, 'mpus': mpus, 'kl': kl, 'z': z, 'octree_out': octree_out}
