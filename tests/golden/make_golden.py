"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py
Imports the reference's python files unmodified (via refenv + the `ocnn`
stand-in) and records inputs / expected outputs.  Only data is written: tensors,
state_dict key names + shapes, sha256 digests.  No reference source is copied.
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refenv  # noqa: E402

refenv.setup()

import torch  # noqa: E402
import common as C  # noqa: E402

from models.networks import modules as RM  # noqa: E402
from models.networks.dualoctree_networks import dual_octree as RD  # noqa: E402
from models.networks.dualoctree_networks import modules as RVM  # noqa: E402
from models.networks.diffusion_networks import graph_unet_hr, graph_unet_lr  # noqa: E402
from models.networks.diffusion_networks.graph_unet_union import UNet3DModel as RUnion  # noqa: E402
from utils.util_dualoctree import split2octree_small, split2octree_large  # noqa: E402

torch.set_grad_enabled(False)


def save(name, obj):
    path = os.path.join(HERE, name + '.pt')
    torch.save(obj, path)
    print('%-28s %8.1f KB' % (name, os.path.getsize(path) / 1024))


def load_filled(module):
    ks = [(k, tuple(v.shape)) for k, v in module.state_dict().items()]
    module.load_state_dict(C.fill_state_dict(ks), strict=True)
    return ks


def octree_record(oc):
    d = oc.depth
    return {'depth': d, 'full_depth': oc.full_depth, 'batch_size': oc.batch_size,
            'nnum': oc.nnum.clone(), 'nnum_nempty': oc.nnum_nempty.clone(),
            'keys': [k.clone() if k is not None else None for k in oc.keys[:d + 1]],
            'children': [c.clone() if c is not None else None for c in oc.children[:d + 1]]}


def canon(edge_idx, edge_dir):
    row, col = edge_idx[0], edge_idx[1]
    n = int(max(int(row.max()), int(col.max())) + 1)
    order = torch.argsort((row * 7 + edge_dir) * n + col, stable=True)
    return torch.stack([row[order], col[order]]), edge_dir[order]


def doctree_record(doc, full=True):
    rec = {'nnum': doc.nnum.clone(), 'lnum': doc.lnum.clone(), 'ncum': doc.ncum.clone(),
           'total_num': doc.total_num, 'depth': doc.depth, 'full_depth': doc.full_depth,
           'batch_size': doc.batch_size, 'graph': {}}
    for d in range(doc.full_depth, doc.depth + 1):
        g = doc.graph[d]
        ei, ed = canon(g['edge_idx'], g['edge_dir'])
        r = {'E': int(ed.numel()), 'N': int(g['node_type'].numel()),
             'sha_edge_idx': C.sha_int(ei), 'sha_edge_dir': C.sha_int(ed),
             'sha_node_type': C.sha_int(g['node_type']), 'sha_keyd': C.sha_int(g['keyd']),
             'sha_node_mask': C.sha_int(g['node_mask']), 'sha_batch_id': C.sha_int(doc.batch_id(d)),
             'sorted_ok': bool((torch.diff(g['edge_idx'][0] * 7 + g['edge_dir']) >= 0).all())}
        if full:
            r.update(edge_idx=ei.to(torch.int32), edge_dir=ed.to(torch.int8),
                     node_type=g['node_type'].to(torch.int8), keyd=g['keyd'].clone(),
                     node_mask=g['node_mask'].clone(), batch_id=doc.batch_id(d).to(torch.int16))
        rec['graph'][d] = r
    return rec


def tiny_doctree(B=2, seed=3):
    split = C.random_split_small(B, 2, seed)
    oc = split2octree_small(split, 4, 2)
    doc = RD.DualOctree(oc)
    doc.post_processing_for_docnn()
    return split, oc, doc


def small_doctree(B=2, seed=4):
    """full_depth 3 / depth 5 tree for the whole-network goldens (better conditioned than 4^3)."""
    split = C.random_split_small(B, 3, seed)
    oc = split2octree_small(split, 5, 3)
    doc = RD.DualOctree(oc)
    doc.post_processing_for_docnn()
    return split, oc, doc


# ---------------------------------------------------------------- G2 / G3
def g_octree_graph():
    split, oc, doc = tiny_doctree()
    out = {'tiny': {'split_small': split, 'octree': octree_record(oc), 'doctree': doctree_record(doc)}}
    # depth-6 -> depth-8 growth on the tiny tree too (full_depth 2 -> 4 -> 6)
    x, y, z, b = oc.xyzb(4)
    sl = C.random_split_large(int(oc.nnum[4]), 11)
    oc_l = split2octree_large(oc, sl, 4)
    doc_l = RD.DualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    out['tiny_large'] = {'split_large': sl, 'octree': octree_record(oc_l),
                         'doctree': doctree_record(doc_l, full=False)}
    # config-shaped: shell-6, B=2 (jittered), and shell-8 B=1 as digests
    sp = C.shell6_split(2, jitter=True)
    oc6 = split2octree_small(sp, 6, 4)
    doc6 = RD.DualOctree(oc6)
    doc6.post_processing_for_docnn()
    rec6 = octree_record(oc6)
    out['shell6_b2'] = {'octree_nnum': rec6['nnum'], 'octree_nnum_nempty': rec6['nnum_nempty'],
                        'sha_keys': [C.sha_int(k) for k in rec6['keys']],
                        'sha_children': [C.sha_int(c) for c in rec6['children']],
                        'doctree': doctree_record(doc6, full=False)}
    sp1 = C.shell6_split(1)
    oc61 = split2octree_small(sp1, 6, 4)
    x, y, z, b = oc61.xyzb(6)
    sl8 = C.shell8_split_large(x, y, z)
    oc8 = split2octree_large(oc61, sl8, 6)
    doc8 = RD.DualOctree(oc8)
    doc8.post_processing_for_docnn()
    rec8 = octree_record(oc8)
    out['shell8_b1'] = {'octree_nnum': rec8['nnum'], 'octree_nnum_nempty': rec8['nnum_nempty'],
                        'sha_keys': [C.sha_int(k) for k in rec8['keys']],
                        'sha_children': [C.sha_int(c) for c in rec8['children']],
                        'doctree': doctree_record(doc8, full=False)}
    save('g_octree_graph', out)


# ---------------------------------------------------------------- G4
def g_modules():
    split, oc, doc = tiny_doctree()
    out = {'split_small': split}
    g = torch.Generator().manual_seed(5)
    N4 = doc.graph[4]['node_type'].numel()
    N3 = doc.graph[3]['node_type'].numel()

    def rnd(name, *s):
        return C.rand_input(name, *s)

    # scatter_mean
    from models.networks.diffusion_networks.utils.scatter import scatter_mean
    src = rnd('sm', 50, 6)
    idx = torch.randint(0, 9, (50,), generator=g)
    out['scatter_mean'] = {'index': idx, 'dim_size': 12,
                           'out': scatter_mean(src, idx, dim=0, dim_size=12)}
    # GraphConv
    for name, cin, cout, nt, bias, d in [('gc_nt0', 5, 7, 0, False, 4), ('gc_nt3_bias', 6, 9, 3, True, 4),
                                         ('gc_d3', 16, 12, 2, False, 3), ('gc_c64', 64, 40, 3, True, 3)]:
        m = RM.GraphConv(cin, cout, 7, 7, nt, use_bias=bias)
        ks = load_filled(m)
        x = rnd(name, doc.graph[d]['node_type'].numel(), cin)
        out[name] = {'d': d, 'args': (cin, cout, 7, 7, nt, bias), 'keys': ks, 'out': m(x, doc, d)}
    # DualOctreeGroupNorm
    for name, c in [('gn12', 12), ('gn64', 64), ('gn60', 60), ('gn96', 96)]:
        m = RM.DualOctreeGroupNorm(c)
        ks = load_filled(m)
        x = rnd(name, N4, c) * 2 + 0.5
        out[name] = {'d': 4, 'c': c, 'group': m.group, 'keys': ks,
                     'out': m(data=x, doctree=doc, depth=4)}
    # Downsample / Upsample
    m = RM.Downsample(6)
    ks = load_filled(m)
    x = rnd('down', 8 * 5, 6)
    out['down'] = {'keys': ks, 'out': m(x)}
    m = RM.Upsample(6)
    ks = load_filled(m)
    x = rnd('up', 5, 6)
    out['up'] = {'keys': ks, 'out': m(x)}
    # GraphDownsample / GraphUpsample (U-Net flavour)
    m = RM.GraphDownsample(8, 10, 7, 7, 2)
    ks = load_filled(m)
    x = rnd('gdown', N4, 8)
    out['gdown'] = {'d': 4, 'args': (8, 10, 7, 7, 2), 'keys': ks, 'out': m(x, doc, 4)}
    m = RM.GraphUpsample(8, 10, 7, 7, 3)
    ks = load_filled(m)
    x = rnd('gup', N3, 8)
    out['gup'] = {'d': 3, 'args': (8, 10, 7, 7, 3), 'keys': ks, 'out': m(x, doc, 3)}
    # VAE flavour
    for name, cin, cout in [('vdown_same', 8, None), ('vdown_diff', 8, 12)]:
        m = RVM.GraphDownsample(cin, cout)
        ks = load_filled(m)
        x = rnd(name, N4, cin)
        lm = doc.node_child(3) < 0
        out[name] = {'args': (cin, cout), 'keys': ks,
                     'out': m(x, doc, 3, lm, int(doc.nnum[4]), int(doc.lnum[3]))}
    for name, cin, cout in [('vup_same', 8, None), ('vup_diff', 8, 12)]:
        m = RVM.GraphUpsample(cin, cout)
        ks = load_filled(m)
        x = rnd(name, N3, cin)
        lm = doc.node_child(3) < 0
        out[name] = {'args': (cin, cout), 'keys': ks,
                     'out': m(x, doc, 4, lm, int(doc.nnum[3]))}
    # GraphResBlockEmbed (B=2, Cin != Cout and Cin == Cout)
    for name, cin, cout in [('rbe_diff', 16, 24), ('rbe_same', 16, None)]:
        m = RM.GraphResBlockEmbed(cin, 32, 0.0, cout, 7, 7, 3)
        ks = load_filled(m)
        x = rnd(name, N4, cin)
        emb = rnd(name + 'e', 2, 32)
        out[name] = {'d': 4, 'args': (cin, 32, 0.0, cout, 7, 7, 3), 'keys': ks,
                     'out': m(x, emb, doc, 4)}
    # GraphResBlocks (VAE)
    m = RM.GraphResBlocks(8, 12, 0.0, 2, 7, 7, 3)
    ks = load_filled(m)
    x = rnd('resblocks', N4, 8)
    out['resblocks'] = {'d': 4, 'args': (8, 12, 0.0, 2, 7, 7, 3), 'keys': ks,
                        'out': m(x, doc, 4)}
    # Conv1x1GnGeluSequential
    m = RM.Conv1x1GnGeluSequential(8, 32)
    ks = load_filled(m)
    x = rnd('c1x1gngelu', N4, 8)
    out['c1x1gngelu'] = {'d': 4, 'keys': ks, 'out': m((x, doc, 4))}
    save('g_modules', out)


# ---------------------------------------------------------------- G5
def g_dense():
    g = torch.Generator().manual_seed(7)
    out = {}
    m = RM.AttentionBlock(32, num_heads=4)
    ks = load_filled(m)
    x = torch.randn(2, 32, 4, 4, 4, generator=g)
    out['attn'] = {'x': x, 'keys': ks, 'out': m(x)}
    m = RM.AttentionBlock(128, num_heads=4)
    ks = load_filled(m)
    x = torch.randn(1, 128, 8, 8, 8, generator=g)
    out['attn512'] = {'x': x, 'keys': ks, 'out': m(x)}
    m = RM.ResnetBlock(3, 8, 12, emb_dim=16, dropout=0.0, use_text_condition=False)
    ks = load_filled(m)
    x = torch.randn(2, 8, 4, 4, 4, generator=g)
    emb = torch.randn(2, 16, generator=g)
    out['resnet'] = {'x': x, 'emb': emb, 'keys': ks, 'out': m(x, emb)}
    # lr net (shrunken), stand-alone and as-middle on the tiny doctree
    split, oc, doc = small_doctree()
    lr = graph_unet_lr.UNet3DModel(**C.TINY_LR_CFG)
    ks = load_filled(lr)
    x = torch.randn(2, 8, 8, 8, 8, generator=g)
    sc = torch.randn(2, 8, 8, 8, 8, generator=g)
    t = torch.tensor([0.3, -1.2])
    out['lr'] = {'x': x, 'x_self_cond': sc, 't': t, 'keys': ks,
                 'out': lr(x=x, timesteps=t, x_self_cond=sc),
                 'out_nosc': lr(x=x, timesteps=t)}
    h = torch.randn(int(doc.nnum[3]), 16, generator=g)
    out['lr_mid'] = {'h': h, 't': t, 'out': lr.forward_as_middle(h, doc, t, None, None)}
    out['split_small'] = split
    save('g_dense', out)


# ---------------------------------------------------------------- G6
def union_cfg(num_classes=None):
    h, l = C.TINY_HR_CFG, C.TINY_LR_CFG
    cfg = dict(stage_flag='hr', image_size=[8, 32], input_depth=[3, 5], unet_type=['lr', 'hr'],
               full_depth=3, input_channels=[8, 3], out_channels=[8, 3],
               model_channels=[l['model_channels'], h['model_channels']],
               num_res_blocks=[[1, 1, 1], h['num_res_blocks']], attention_resolutions=[2, 4],
               channel_mult=[l['channel_mult'], h['channel_mult']], num_heads=4,
               use_checkpoint=False, dims=3)
    if num_classes is not None:
        cfg['num_classes'] = num_classes
    return cfg


def g_unet():
    g = torch.Generator().manual_seed(9)
    split, oc, doc = small_doctree()
    out = {'split_small': split}
    for name, ncls in [('uncond', None), ('cond', 5)]:
        net = RUnion(**union_cfg(ncls)).eval()
        ks = load_filled(net)
        x = torch.randn(doc.total_num, 3, generator=g)
        t = torch.tensor([0.7, 0.7])
        label = torch.tensor([1, 3]) if ncls else None
        y = net(unet_type='hr', x=x, doctree=doc, unet_lr=net.unet_lr, timesteps=t,
                x_self_cond=None, label=label)
        out[name] = {'x': x, 't': t, 'label': label, 'keys': ks, 'out': y, 'num_classes': ncls}
    # 3-stage nesting (feature stage nests hr as_middle without lr): fd 2, depths 2/4/6
    sl = C.random_split_large(int(oc.nnum[5]), 11, p=0.25)
    oc_l = split2octree_large(oc, sl, 5)
    doc_l = RD.DualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    cfg3 = dict(stage_flag='feature', image_size=[8, 32, 128], input_depth=[3, 5, 7],
                unet_type=['lr', 'hr', 'feature'], full_depth=3, input_channels=[8, 8, 3],
                out_channels=[8, 8, 3], model_channels=[16, 32, 32],
                num_res_blocks=[[1, 1, 1], [1, 1, 0], [1, 1, 1]], attention_resolutions=[2, 4],
                channel_mult=[[1, 2, 4], [1, 2, 4], [1, 2, 4]], num_heads=4,
                use_checkpoint=False, dims=3)
    net = RUnion(**cfg3).eval()
    ks = load_filled(net)
    x = torch.randn(doc_l.total_num, 3, generator=g)
    t = torch.tensor([-0.4, -0.4])
    y = net(unet_type='feature', x=x, doctree=doc_l, unet_lr=net.unet_hr, timesteps=t,
            x_self_cond=None, label=None)
    out['feature'] = {'x': x, 't': t, 'keys': ks, 'out': y, 'split_large': sl, 'cfg': cfg3}
    save('g_unet', out)


# ---------------------------------------------------------------- G7
def g_sample_loop():
    from models import octfusion_model_union as OM
    from models.networks.diffusion_networks.ldm_diffusion_util import beta_linear_log_snr
    out = {}

    class Fake:
        pass

    def run(shape, unet_type, df_type, trunc, B, steps, seed):
        fs = Fake()
        fs.device = 'cpu'
        fs.log_snr = beta_linear_log_snr
        fs.vq_conf = refenv._AttrDict({'data': {'test': {'batch_size': B}}})
        fs.get_sampling_timesteps = types.MethodType(OM.OctFusionModel.get_sampling_timesteps, fs)
        A = torch.linspace(-0.5, 0.5, shape[-1] if len(shape) == 2 else shape[1])

        def net(unet_type=None, x=None, doctree=None, timesteps=None, unet_lr=None,
                x_self_cond=None, label=None):
            # deterministic fake denoiser: depends on x, the noise level and the self-cond
            a = A.view(1, -1, *([1] * (x.ndim - 2)))
            tt = timesteps.view(-1, *([1] * (x.ndim - 1))) if x.ndim > 2 else timesteps[0]
            y = torch.tanh(x * 0.7 + a) * 0.9 + 0.05 * torch.tanh(tt)
            if x_self_cond is not None:
                y = y + 0.1 * x_self_cond
            return y
        fs.ema_df = net
        fs.df = net
        torch.manual_seed(seed)
        res = OM.OctFusionModel.sample_loop.__wrapped__(
            fs, doctree_lr=None, ema=True, shape=shape, ddim_steps=steps, label=None,
            unet_type=unet_type, unet_lr=None, df_type=df_type, truncated_index=trunc)
        return res

    out['x0'] = {'shape': (2, 8, 4, 4, 4), 'B': 2, 'steps': 6, 'seed': 123, 'trunc': 0.7,
                 'out': run((2, 8, 4, 4, 4), 'lr', 'x0', 0.7, 2, 6, 123)}
    out['eps'] = {'shape': (37, 3), 'B': 2, 'steps': 6, 'seed': 321,
                  'out': run((37, 3), 'hr', 'eps', 0.0, 2, 6, 321)}
    out['x0_graph'] = {'shape': (37, 3), 'B': 1, 'steps': 5, 'seed': 77,
                       'out': run((37, 3), 'hr', 'x0', 0.0, 1, 5, 77)}
    save('g_sample_loop', out)


# ---------------------------------------------------------------- G8
VAE_CFG = dict(depth=6, channel_in=4, nout=4, full_depth=2, depth_stop=4, depth_out=6, use_checkpoint=False,
               resblk_type='basic', bottleneck=4, resblk_num=2, code_channel=16, embed_dim=3)


def g_vae():
    from models.networks.dualoctree_networks.graph_vae import GraphVAE
    split, oc, doc = tiny_doctree()
    sl = C.random_split_large(int(oc.nnum[4]), 11, p=0.3)
    oc_l = split2octree_large(oc, sl, 4)
    doc_l = RD.DualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    vae = GraphVAE(**VAE_CFG).eval()
    ks = load_filled(vae)
    N4 = doc_l.graph[4]['node_type'].numel()
    code = C.rand_input('vae_code', N4, 3)
    out = vae.decode_code(code, doc_l, update_octree=False)
    rec = {'split_small': split, 'split_large': sl, 'keys': ks, 'cfg': VAE_CFG,
           'logits': {d: v.clone() for d, v in out['logits'].items()},
           'reg_voxs': {d: v.clone() for d, v in out['reg_voxs'].items()}}
    # update_octree=True: structure depends on argmax decisions; record them with their margins
    import copy
    doc_in = RD.DualOctree(copy.deepcopy(oc_l))
    doc_in.post_processing_for_docnn()
    out2 = vae.decode_code(code, doc_in, update_octree=True)
    rec['grow'] = {'logits': {d: v.clone() for d, v in out2['logits'].items()},
                   'nnum': out2['octree_out'].nnum.clone(), 'nnum_nempty': out2['octree_out'].nnum_nempty.clone(),
                   'reg_shapes': {d: tuple(v.shape) for d, v in out2['reg_voxs'].items()}}
    save('g_vae', rec)


# ---------------------------------------------------------------- G10
def g_vae_enc():
    """GraphVAE encoder (octree_encoder_step + KL_conv) on the tiny depth-6 tree with a random input feature
    (the reference reads it from the ocnn octree: graph_vae.py:131-132; patched to return the tensor)."""
    from models.networks.dualoctree_networks.graph_vae import GraphVAE
    split, oc, doc = tiny_doctree()
    sl = C.random_split_large(int(oc.nnum[4]), 11, p=0.3)
    oc_l = split2octree_large(oc, sl, 4)
    doc_l = RD.DualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    vae = GraphVAE(**VAE_CFG).eval()
    ks = load_filled(vae)
    N6 = doc_l.graph[6]['node_type'].numel()
    data = C.rand_input('vae_enc_in', N6, 4)
    vae._get_input_feature = lambda d: data
    convs = vae.octree_encoder_step(oc_l, doc_l)
    code = vae.KL_conv(convs[4])
    save('g_vae_enc', {'split_small': split, 'split_large': sl, 'keys': ks, 'cfg': VAE_CFG,
                       'h_rows8': convs[4][::8].clone(), 'kl': code.clone()})      # every 8th row of h: small fixture


# ---------------------------------------------------------------- G9
def mpu_points(n, B, seed):
    """Query points: uniform in the cube, plus points on / beyond cell-centre planes and the cube boundary."""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g) * 2 - 1
    p[: n // 8] = (torch.randint(0, 65, (n // 8, 3), generator=g).float() / 32 - 1)      # exact grid planes
    p[n // 8: n // 6] = p[n // 8: n // 6].sign()                                         # cube corners / faces
    b = torch.randint(0, B, (n, 1), generator=g).float()
    return torch.cat([p, b], 1)


def g_mpu():
    """NeuralMPU (mpu.py) on the tiny depth-6 tree of g_vae: reference outputs for random per-node codes."""
    from models.networks.dualoctree_networks import mpu as RMPU
    split, oc, doc = tiny_doctree()
    sl = C.random_split_large(int(oc.nnum[4]), 11, p=0.3)
    oc_l = split2octree_large(oc, sl, 4)
    full_depth, depth_stop, depth = 2, 4, 6
    ncum = torch.cumsum(oc_l.nnum, 0)
    reg = {d: C.rand_input('mpu_code_%d' % d, int(ncum[d] - (ncum[full_depth - 1] if full_depth else 0)), 4)
           for d in range(depth_stop, depth + 1)}
    pos = mpu_points(4096, oc_l.batch_size, 21)
    cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self        # mpu.py:128 hard-codes .cuda(); no GPU here
    try:
        out = RMPU.NeuralMPU(full_depth, depth_stop, depth)(pos, reg, oc_l)
    finally:
        torch.Tensor.cuda = cuda
    save('g_mpu', {'split_small': split, 'split_large': sl, 'pos': pos,
                   'cfg': (full_depth, depth_stop, depth),
                   'sdf': {d: v[0].clone() for d, v in out.items()},
                   'mask': {d: v[1].clone() for d, v in out.items()}})



# ---------------------------------------------------------------- G11
def g_vae_train():
    """One VAE training forward + losses + parameter gradients from the reference's own code:
    GraphVAE.forward(octree_in, octree_gt, pos) (graph_vae.py:246-289) -> loss.geometry_loss (loss.py:164-178,
    'sdf_reg_loss', kl_weight 0.1 as configs/vae_snet_train.yaml:92-95) -> autograd.  The input feature and the
    posterior noise are given tensors (the reference reads the former from the ocnn octree and draws the latter
    with torch.randn: both patched to return the recorded inputs).  Re-running reproduces every other fixture of this
    directory bit for bit and this one's losses exactly; its parameter gradients come back within 1e-6 of the tensor's
    range (the summation order of CPU autograd's threaded index_add is not fixed) -- the tests allow 5e-3."""
    import copy
    from models.networks.dualoctree_networks.graph_vae import GraphVAE
    from models.networks.dualoctree_networks import loss as RL
    split, oc, doc = tiny_doctree()
    sl = C.random_split_large(int(oc.nnum[4]), 11, p=0.3)
    oc_l = split2octree_large(oc, sl, 4)
    oc_gt = copy.deepcopy(oc_l)
    vae = GraphVAE(**VAE_CFG).train()
    ks = load_filled(vae)
    doc_l = RD.DualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    N6 = doc_l.graph[6]['node_type'].numel()
    N4 = doc_l.graph[4]['node_type'].numel()
    data = C.rand_input('vae_enc_in', N6, 4)
    noise = C.rand_input('vae_post_noise', N4, VAE_CFG['embed_dim'])
    vae._get_input_feature = lambda d: data
    n_pts = 2048
    pos = mpu_points(n_pts, oc_l.batch_size, 33)
    sdf_gt = C.rand_input('vae_sdf_gt', n_pts, 1).view(-1) * 0.05
    grad_gt = torch.nn.functional.normalize(C.rand_input('vae_grad_gt', n_pts, 3), dim=1)
    batch = {'pos': pos.clone().requires_grad_(True), 'sdf': sdf_gt, 'grad': grad_gt}
    randn, cuda = torch.randn, torch.Tensor.cuda

    def fixed_randn(*shape, **kw):
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert shp == tuple(noise.shape), shp
        return noise.clone()
    torch.randn = fixed_randn
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with torch.enable_grad():
            out = vae(oc_l, oc_gt, batch['pos'])
            mpu_grad = {d: g.detach().clone() for d, g in RL.compute_mpu_gradients(out['mpus'], batch['pos']).items()}
            losses = RL.geometry_loss(batch, out, 'sdf_reg_loss', kl_weight=0.1)
            total = torch.sum(torch.stack([v for k, v in losses.items() if 'loss' in k]))   # octfusion_model_vae.py:182-183
            for t in list(out['logits'].values()) + list(out['reg_voxs'].values()):
                t.retain_grad()
            total.backward()
    finally:
        torch.randn = randn
        torch.Tensor.cuda = cuda
    def sub(v, cap=2048):
        """strided sample of a tensor + its L2 norm (keeps the fixture small; every element still moves the norm)"""
        f = v.detach().reshape(-1)
        stride = max(1, -(-f.numel() // cap))
        return {'stride': stride, 'vals': f[::stride].clone(), 'norm': float(f.double().norm()), 'shape': tuple(v.shape)}
    grads = {k: sub(p.grad) if p.grad is not None else None for k, p in vae.named_parameters()}
    rec = {'split_small': split, 'split_large': sl, 'keys': ks, 'cfg': VAE_CFG, 'pos': pos, 'sdf_gt': sdf_gt,
           'grad_gt': grad_gt, 'kl_weight': 0.1, 'n_noise': tuple(noise.shape),
           'losses': {k: float(v) for k, v in losses.items()}, 'total': float(total),
           'logits': {d: sub(v) for d, v in out['logits'].items()},
           'reg_voxs': {d: sub(v) for d, v in out['reg_voxs'].items()},
           'd_logits': {d: sub(v.grad) for d, v in out['logits'].items()},
           'd_reg_voxs': {d: sub(v.grad) for d, v in out['reg_voxs'].items()},
           'sdf': {d: v[0].detach().clone() for d, v in out['mpus'].items()},
           'mpu_grad': mpu_grad,
           'grads': grads}
    save('g_vae_train', rec)

# ---------------------------------------------------------------- G12: the drop-in boundary at the REAL configs
def g_boundary():
    """What `load_ckpt(strict=True)` binds (octfusion_model_union.py:525-545, model_utils.py:18-28): the state_dict
    key / shape lists, in order, of the reference's own nets built from its three diffusion YAMLs and its ShapeNet /
    Objaverse VAE YAMLs -- plus the reference-signature dense blocks (modules.py:63-95, 474-547: [b, c, D, H, W] in
    and out) on small inputs.  Keys and shapes go to a JSON file (readable diffs), tensors to g_boundary.pt."""
    import json
    from models.networks.dualoctree_networks.graph_vae import GraphVAE
    cfg_dir = os.path.join(refenv.REF, 'configs')
    rec = {}
    for name, yaml_name, stage in (('snet_uncond', 'octfusion_snet_uncond.yaml', 'hr'),
                                   ('snet_cond', 'octfusion_snet_cond.yaml', 'hr'),
                                   ('obja_uncond', 'octfusion_obja_uncond.yaml', 'feature')):
        cfg = refenv.load_yaml(os.path.join(cfg_dir, yaml_name))
        params = dict(cfg['unet']['params'])
        params.pop('df_type', None)
        with torch.device('meta'):
            net = RUnion(stage, **params)
        rec[name] = {'yaml': yaml_name, 'stage_flag': stage,
                     'keys': [[k, list(v.shape)] for k, v in net.state_dict().items()]}
    for name, yaml_name in (('vae_snet', 'vae_snet_eval.yaml'), ('vae_obja_depth864', 'vae_obja_eval_depth864.yaml')):
        m = refenv.load_yaml(os.path.join(cfg_dir, yaml_name))['model']
        with torch.device('meta'):
            vae = GraphVAE(depth=m['depth'], channel_in=m['channel'], nout=m['nout'], full_depth=m['full_depth'],
                           depth_stop=m['depth_stop'], depth_out=m['depth_out'], use_checkpoint=False,
                           resblk_type=m['resblock_type'], bottleneck=m['bottleneck'], resblk_num=m['resblk_num'],
                           code_channel=m['code_channel'], embed_dim=m['embed_dim'])
        rec[name] = {'yaml': yaml_name, 'model': {k: m[k] for k in ('depth', 'channel', 'nout', 'full_depth', 'depth_stop',
                                                                    'depth_out', 'resblock_type', 'bottleneck',
                                                                    'resblk_num', 'code_channel', 'embed_dim')},
                     'keys': [[k, list(v.shape)] for k, v in vae.state_dict().items()]}
    with open(os.path.join(HERE, 'boundary_keys.json'), 'w') as f:
        json.dump(rec, f, indent=0)
    print('%-28s %8.1f KB' % ('boundary_keys.json', os.path.getsize(os.path.join(HERE, 'boundary_keys.json')) / 1024))
    # dense blocks through their reference signatures
    g = torch.Generator().manual_seed(21)
    out = {}
    m = RM.ConvDownsample(16, dims=3)
    ks = load_filled(m)
    x = torch.randn(2, 16, 8, 8, 8, generator=g)
    out['convdown'] = {'x': x, 'keys': ks, 'out': m(x)}
    m = RM.ConvUpsample(16, dims=3)
    ks = load_filled(m)
    x = torch.randn(2, 16, 4, 4, 4, generator=g)
    out['convup'] = {'x': x, 'keys': ks, 'out': m(x)}
    m = RM.ResnetBlock(3, 32, 32, emb_dim=16, dropout=0.0, use_text_condition=False)
    ks = load_filled(m)
    x = torch.randn(2, 32, 8, 8, 8, generator=g)
    emb = torch.randn(2, 16, generator=g)
    out['resnet_same'] = {'x': x, 'emb': emb, 'keys': ks, 'out': m(x, emb)}
    save('g_boundary', out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['octree_graph', 'modules', 'dense', 'unet', 'sample_loop', 'vae', 'mpu', 'vae_enc', 'vae_train',
                             'boundary']
    for w in which:
        globals()['g_' + w]()
