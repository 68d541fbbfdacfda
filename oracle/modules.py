"""Oracle: building blocks of the dual-octree U-Net (torch CPU fp32).  TEST INFRASTRUCTURE.

Functional restatement of reference models/networks/modules.py and
.../utils/scatter.py, in the reference's own op sequence (so timing it is a
fair CPU baseline): ``x[col]`` index -> zeros -> scatter_add_ (values) ->
scatter_add_ (ones) -> clamp/divide -> mm.  Parameters are passed explicitly
(tensors / state_dict slices with the reference's key names).
"""
import math

import torch
import torch.nn.functional as F

# Working float type of the spots where the reference says `.float()` (GroupNorm32, modules.py:26-28; the softmax of
# QKVAttention, :546; the timestep embedding).  float32 = the reference.  Tests that need an error figure for an
# fp32-class implementation below the fp32 reference's own rounding noise run the SAME op sequence in float64
# (`with oracle.modules.working_float(torch.float64)`, all tensors passed as doubles).
FLOAT = torch.float32


class working_float:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global FLOAT
        self.saved, FLOAT = FLOAT, self.dtype

    def __exit__(self, *exc):
        global FLOAT
        FLOAT = self.saved


from .octree import scatter_add


def scatter_mean(src, index, dim_size):
    """.../utils/scatter.py:42-66 (weights=None, dim=0)."""
    out = scatter_add(src, index, dim=0, dim_size=dim_size)
    ones = torch.ones(index.size(), dtype=src.dtype)
    count = scatter_add(ones, index, dim=0, dim_size=dim_size)
    count[count < 1] = 1
    out.true_divide_(count.unsqueeze(-1))
    return out


def graph_conv(x, doctree, d, weights, bias=None, n_node_type=0, n_edge_type=7):
    """modules.py:194-220 GraphConv.forward."""
    g = doctree.graph[d]
    edge_idx, edge_dir, node_type = g['edge_idx'], g['edge_dir'], g['node_type']
    if node_type is not None and n_node_type > 1:
        one_hot = F.one_hot(node_type, num_classes=n_node_type)
        x = torch.cat([x, one_hot], dim=1)
    row, col = edge_idx[0], edge_idx[1]
    index = row * n_edge_type + edge_dir
    col_data = scatter_mean(x[col], index, x.shape[0] * n_edge_type)
    out = col_data.view(x.shape[0], -1) @ weights
    if bias is not None:
        out += bias
    return out


def gn_groups(channels, group=32):
    """modules.py:271-280 group-count rule."""
    if channels <= 32:
        group = channels // 4
    elif channels % group != 0:
        group = 30
    assert channels % group == 0
    return group


def dual_octree_group_norm(data, doctree, depth, weights, bias, group=32, eps=1e-5):
    """modules.py:291-326 DualOctreeGroupNorm.forward."""
    C = data.shape[1]
    group = gn_groups(C, group)
    cpg = C // group
    B = doctree.batch_size
    batch_id = doctree.batch_id(depth)
    assert batch_id.shape[0] == data.shape[0]

    def adjust(t):
        if cpg > 1:
            t = t.reshape(-1, group, cpg).sum(-1, keepdim=True).repeat(1, 1, cpg).reshape(-1, C)
        return t

    ones = data.new_ones([data.shape[0], 1])
    count = scatter_add(ones, batch_id, dim=0, dim_size=B) * cpg
    inv_count = 1.0 / (count + eps)
    mean = adjust(scatter_add(data, batch_id, dim=0, dim_size=B) * inv_count)
    out = data - mean.index_select(0, batch_id)
    var = adjust(scatter_add(out ** 2, batch_id, dim=0, dim_size=B) * inv_count)
    inv_std = 1.0 / (var + eps).sqrt()
    out = out * inv_std.index_select(0, batch_id)
    return out * weights + bias


def downsample(x, weights):
    """modules.py:391-395: [n*8, C] -> [n, C] with W [C, C, 8]."""
    C = weights.shape[0]
    return x.view(-1, C * 8) @ weights.flatten(1).t()


def upsample(x, weights):
    """modules.py:440-443: [n, C] -> [n*8, C]."""
    C = weights.shape[0]
    return (x @ weights.flatten(1)).view(-1, C)


def pool_rearrange(x, doctree, d, down_w):
    """modules.py:409-423 (GraphDownsample up to, not including, the conv)."""
    numd = int(doctree.nnum[d])
    lnumd = int(doctree.lnum[d - 1])
    leaf_mask = doctree.node_child(d - 1) < 0
    outd = downsample(x[-numd:], down_w)
    out = torch.zeros(leaf_mask.shape[0], x.shape[1], dtype=x.dtype)
    out[leaf_mask] = x[-lnumd - numd:-numd]
    out[leaf_mask.logical_not()] = outd
    return torch.cat([x[:-numd - lnumd], out], dim=0)


def unpool_rearrange(x, doctree, d, up_w):
    """modules.py:458-467 (GraphUpsample up to, not including, the conv)."""
    numd = int(doctree.nnum[d])
    leaf_mask = doctree.node_child(d) < 0
    outd = x[-numd:]
    out1 = upsample(outd[leaf_mask.logical_not()], up_w)
    return torch.cat([x[:-numd], outd[leaf_mask], out1], dim=0)


def _sub(sd, prefix):
    p = prefix + '.'
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


def graph_downsample(x, doctree, d, sd, n_node_type):
    """modules.py:400-428 (U-Net flavour): pool d -> d-1 then GraphConv at d-1."""
    out = pool_rearrange(x, doctree, d, sd['downsample.weights'])
    return graph_conv(out, doctree, d - 1, sd['conv.weights'], None, n_node_type)


def graph_upsample(x, doctree, d, sd, n_node_type):
    """modules.py:449-472 (U-Net flavour): unpool d -> d+1 then GraphConv at d+1."""
    out = unpool_rearrange(x, doctree, d, sd['upsample.weights'])
    return graph_conv(out, doctree, d + 1, sd['conv.weights'], None, n_node_type)


def silu(x):
    """nn.SiLU (modules.py:42-43, 706; graph_unet_hr.py:109,205)."""
    return F.silu(x)


def swish(x):
    """modules.py:30-32 `nonlinearity` (VAE res-blocks)."""
    return x * torch.sigmoid(x)


def graph_resblock_embed(x, emb, doctree, depth, sd, n_node_type):
    """modules.py:730-763 GraphResBlockEmbed._forward (use_scale_shift_norm=False)."""
    h = dual_octree_group_norm(x, doctree, depth, sd['block1_norm.weights'], sd['block1_norm.bias'])
    h = silu(h)
    h = graph_conv(h, doctree, depth, sd['conv1.weights'], None, n_node_type)
    emb_out = F.linear(silu(emb), sd['emb_layers.1.weight'], sd['emb_layers.1.bias'])
    batch_id = doctree.batch_id(depth)
    assert doctree.batch_size == emb_out.shape[0]
    for i in range(doctree.batch_size):          # the reference's O(B) masked-add loop
        h[batch_id == i] += emb_out[i]
    h = dual_octree_group_norm(h, doctree, depth, sd['block2_norm.weights'], sd['block2_norm.bias'])
    h = silu(h)
    h = graph_conv(h, doctree, depth, sd['conv2.weights'], None, n_node_type)
    if 'skip_connection.linear.weight' in sd:
        x = F.linear(x, sd['skip_connection.linear.weight'], sd.get('skip_connection.linear.bias'))
    return x + h


def conv1x1_gn(x, doctree, depth, sd, gelu=False):
    """modules.py:341-365 Conv1x1Gn / Conv1x1GnGelu."""
    out = F.linear(x, sd['conv.linear.weight'])
    out = dual_octree_group_norm(out, doctree, depth, sd['gn.weights'], sd['gn.bias'])
    return F.gelu(out) if gelu else out


def graph_resblock(x, doctree, depth, sd, n_node_type):
    """modules.py:625-641 GraphResBlock._forward (VAE flavour, no embedding)."""
    h = dual_octree_group_norm(x, doctree, depth, sd['norm1.weights'], sd['norm1.bias'])
    h = graph_conv(swish(h), doctree, depth, sd['conv1.weights'], None, n_node_type)
    h = dual_octree_group_norm(h, doctree, depth, sd['norm2.weights'], sd['norm2.bias'])
    h = graph_conv(swish(h), doctree, depth, sd['conv2.weights'], None, n_node_type)
    if 'conv1x1c.conv.linear.weight' in sd:
        x = conv1x1_gn(x, doctree, depth, _sub(sd, 'conv1x1c'))
    return h + x


def graph_resblocks(x, doctree, depth, sd, n_node_type):
    """modules.py:643-659 GraphResBlocks."""
    i = 0
    while any(k.startswith('resblks.%d.' % i) for k in sd):
        x = graph_resblock(x, doctree, depth, _sub(sd, 'resblks.%d' % i), n_node_type)
        i += 1
    return x


# ---- dense (16^3 voxel) blocks: modules.py:26-95, 474-563 -------------------

def group_norm32(x, w, b, channels):
    return F.group_norm(x.to(FLOAT), min(channels, 32), w, b, 1e-5)


def resnet_block(x, emb, sd):
    """modules.py:505-513 ResnetBlock.forward (use_text_condition=False)."""
    cin = x.shape[1]
    h = group_norm32(x, sd['block1.0.weight'], sd['block1.0.bias'], cin)
    h = F.conv3d(silu(h), sd['block1.2.weight'], sd['block1.2.bias'], padding=1)
    t = F.linear(silu(emb), sd['time_mlp.1.weight'], sd['time_mlp.1.bias'])
    h = h + t[:, :, None, None, None]
    cout = h.shape[1]
    h = group_norm32(h, sd['block2.0.weight'], sd['block2.0.bias'], cout)
    h = F.conv3d(silu(h), sd['block2.3.weight'], sd['block2.3.bias'], padding=1)
    if 'res_conv.weight' in sd:
        x = F.conv3d(x, sd['res_conv.weight'], sd['res_conv.bias'])
    return h + x


def attention_block(x, sd, num_heads):
    """modules.py:527-547 AttentionBlock.forward + QKVAttention."""
    b, c = x.shape[:2]
    spatial = x.shape[2:]
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(x, sd['norm.weight'], sd['norm.bias'], c),
                   sd['qkv.weight'], sd['qkv.bias'])
    qkv = qkv.reshape(b * num_heads, -1, qkv.shape[2])
    ch = qkv.shape[1] // 3
    q, k, v = torch.split(qkv, ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum('bct,bcs->bts', q * scale, k * scale)
    w = torch.softmax(w.to(FLOAT), dim=-1)
    h = torch.einsum('bts,bcs->bct', w, v).reshape(b, -1, x.shape[-1])
    h = F.conv1d(h, sd['proj_out.weight'], sd['proj_out.bias'])
    return (x + h).reshape(b, c, *spatial)


def attn_seq(x, sd, num_heads):
    """graph_unet_lr.py:128-132: Sequential(GN32, SiLU, AttentionBlock)."""
    c = x.shape[1]
    h = silu(group_norm32(x, sd['0.weight'], sd['0.bias'], c))
    return attention_block(h, _sub(sd, '2'), num_heads)
