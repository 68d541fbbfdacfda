"""The CPU oracle against golden vectors captured from the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import torch
import pytest

import common as C
from oracle import dual_octree as OD
from oracle import modules as OM
from oracle import sampler as OS
from oracle import unet as OU
from oracle.octree import key2xyz, xyz2key

torch.set_grad_enabled(False)
TOL = dict(rtol=1e-4, atol=1e-5)


def tiny(split):
    oc = OS.split2octree_small(split, 4, 2)
    doc = OD.OracleDualOctree(oc)
    doc.post_processing_for_docnn()
    return oc, doc


def small(split):
    oc = OS.split2octree_small(split, 5, 3)
    doc = OD.OracleDualOctree(oc)
    doc.post_processing_for_docnn()
    return oc, doc


def check_octree(oc, rec):
    assert oc.depth == rec['depth']
    assert torch.equal(oc.nnum[:oc.depth + 1], rec['nnum'][:oc.depth + 1])
    assert torch.equal(oc.nnum_nempty[:oc.depth + 1], rec['nnum_nempty'][:oc.depth + 1])
    for d in range(oc.depth + 1):
        assert torch.equal(oc.keys[d], rec['keys'][d])
        assert torch.equal(oc.children[d].to(torch.int64), rec['children'][d].to(torch.int64))


def check_doctree(doc, rec):
    assert doc.total_num == rec['total_num']
    assert torch.equal(doc.nnum, rec['nnum']) and torch.equal(doc.lnum, rec['lnum'])
    assert torch.equal(doc.ncum, rec['ncum'])
    for d, r in rec['graph'].items():
        g = doc.graph[d]
        ei, ed = OD.canonical_edges(g['edge_idx'], g['edge_dir'])
        assert ed.numel() == r['E'] and g['node_type'].numel() == r['N']
        assert C.sha_int(ei) == r['sha_edge_idx']
        assert C.sha_int(ed) == r['sha_edge_dir']
        assert C.sha_int(g['node_type']) == r['sha_node_type']
        assert C.sha_int(g['keyd']) == r['sha_keyd']
        assert C.sha_int(g['node_mask']) == r['sha_node_mask']
        assert C.sha_int(doc.batch_id(d)) == r['sha_batch_id']
        if 'edge_idx' in r:
            assert torch.equal(ei, r['edge_idx'].to(torch.int64))
            assert torch.equal(ed, r['edge_dir'].to(torch.int64))


def test_key_codec_roundtrip():
    g = torch.Generator().manual_seed(0)
    x, y, z = (torch.randint(0, 256, (1000,), generator=g) for _ in range(3))
    b = torch.randint(0, 7, (1000,), generator=g)
    k = xyz2key(x, y, z, b, depth=8)
    x2, y2, z2, b2 = key2xyz(k, 8)
    assert torch.equal(x, x2) and torch.equal(y, y2) and torch.equal(z, z2) and torch.equal(b, b2)
    # child-octant bit order x->4, y->2, z->1 (dual_octree.py:85-94)
    assert int(xyz2key(torch.tensor([1]), torch.tensor([0]), torch.tensor([0]), depth=1)) == 4
    assert int(xyz2key(torch.tensor([0]), torch.tensor([1]), torch.tensor([0]), depth=1)) == 2
    assert int(xyz2key(torch.tensor([0]), torch.tensor([0]), torch.tensor([1]), depth=1)) == 1


def test_octree_and_graph_tiny(golden):
    G = golden('g_octree_graph')
    t = G['tiny']
    oc, doc = tiny(t['split_small'])
    check_octree(oc, t['octree'])
    check_doctree(doc, t['doctree'])
    tl = G['tiny_large']
    oc_l = OS.split2octree_large(oc, tl['split_large'], 4)
    check_octree(oc_l, tl['octree'])
    doc_l = OD.OracleDualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    check_doctree(doc_l, tl['doctree'])


def test_octree_and_graph_shell(golden):
    G = golden('g_octree_graph')
    s6 = G['shell6_b2']
    oc = OS.split2octree_small(C.shell6_split(2, jitter=True), 6, 4)
    assert torch.equal(oc.nnum, s6['octree_nnum'])
    for d in range(7):
        assert C.sha_int(oc.keys[d]) == s6['sha_keys'][d]
        assert C.sha_int(oc.children[d]) == s6['sha_children'][d]
    doc = OD.OracleDualOctree(oc)
    doc.post_processing_for_docnn()
    check_doctree(doc, s6['doctree'])
    # shell-8 (nnum7 = 80320, nnum8 = 402560, N8 = 448232, E8 = 3374048)
    s8 = G['shell8_b1']
    oc6 = OS.split2octree_small(C.shell6_split(1), 6, 4)
    assert oc6.nnum.tolist() == [1, 8, 64, 512, 4096, 4672, 20032]
    x, y, z, b = oc6.xyzb(6)
    oc8 = OS.split2octree_large(oc6, C.shell8_split_large(x, y, z), 6)
    assert torch.equal(oc8.nnum, s8['octree_nnum'])
    assert int(oc8.nnum[7]) == 80320 and int(oc8.nnum[8]) == 402560
    doc8 = OD.OracleDualOctree(oc8)
    doc8.post_processing_for_docnn()
    check_doctree(doc8, s8['doctree'])
    assert s8['doctree']['graph'][8]['N'] == 448232 and s8['doctree']['graph'][8]['E'] == 3374048


def test_modules(golden):
    G = golden('g_modules')
    oc, doc = tiny(G['split_small'])
    r = G['scatter_mean']
    torch.testing.assert_close(OM.scatter_mean(C.rand_input('sm', 50, 6), r['index'], r['dim_size']),
                               r['out'], **TOL)
    for name in ['gc_nt0', 'gc_nt3_bias', 'gc_d3', 'gc_c64']:
        r = G[name]
        sd = C.fill_state_dict(r['keys'])
        cin, cout, _, _, nt, bias = r['args']
        x = C.rand_input(name, doc.graph[r['d']]['node_type'].numel(), cin)
        y = OM.graph_conv(x, doc, r['d'], sd['weights'], sd.get('bias'), nt)
        torch.testing.assert_close(y, r['out'], **TOL)
    N4 = doc.graph[4]['node_type'].numel()
    N3 = doc.graph[3]['node_type'].numel()
    for name in ['gn12', 'gn64', 'gn60', 'gn96']:
        r = G[name]
        sd = C.fill_state_dict(r['keys'])
        assert OM.gn_groups(r['c']) == r['group']
        x = C.rand_input(name, N4, r['c']) * 2 + 0.5
        torch.testing.assert_close(OM.dual_octree_group_norm(x, doc, 4, sd['weights'], sd['bias']),
                                   r['out'], **TOL)
    sd = C.fill_state_dict(G['down']['keys'])
    torch.testing.assert_close(OM.downsample(C.rand_input('down', 40, 6), sd['weights']), G['down']['out'], **TOL)
    sd = C.fill_state_dict(G['up']['keys'])
    torch.testing.assert_close(OM.upsample(C.rand_input('up', 5, 6), sd['weights']), G['up']['out'], **TOL)
    r = G['gdown']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(OM.graph_downsample(C.rand_input('gdown', N4, 8), doc, 4, sd, 2), r['out'], **TOL)
    r = G['gup']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(OM.graph_upsample(C.rand_input('gup', N3, 8), doc, 3, sd, 3), r['out'], **TOL)
    for name in ['vdown_same', 'vdown_diff']:
        r = G[name]
        sd = C.fill_state_dict(r['keys'])
        y = OM.pool_rearrange(C.rand_input(name, N4, 8), doc, 4, sd['downsample.weights'])
        if r['args'][1]:
            y = OM.conv1x1_gn(y, doc, 3, OM._sub(sd, 'conv1x1'), gelu=True)
        torch.testing.assert_close(y, r['out'], **TOL)
    for name in ['vup_same', 'vup_diff']:
        r = G[name]
        sd = C.fill_state_dict(r['keys'])
        y = OM.unpool_rearrange(C.rand_input(name, N3, 8), doc, 3, sd['upsample.weights'])
        if r['args'][1]:
            y = OM.conv1x1_gn(y, doc, 4, OM._sub(sd, 'conv1x1'), gelu=True)
        torch.testing.assert_close(y, r['out'], **TOL)
    for name in ['rbe_diff', 'rbe_same']:
        r = G[name]
        sd = C.fill_state_dict(r['keys'])
        cin = r['args'][0]
        y = OM.graph_resblock_embed(C.rand_input(name, N4, cin), C.rand_input(name + 'e', 2, 32),
                                    doc, 4, sd, 3)
        torch.testing.assert_close(y, r['out'], **TOL)
    r = G['resblocks']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(OM.graph_resblocks(C.rand_input('resblocks', N4, 8), doc, 4, sd, 3),
                               r['out'], **TOL)
    r = G['c1x1gngelu']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(OM.conv1x1_gn(C.rand_input('c1x1gngelu', N4, 8), doc, 4, sd, gelu=True),
                               r['out'], **TOL)


def test_dense(golden):
    G = golden('g_dense')
    r = G['attn']
    torch.testing.assert_close(OM.attention_block(r['x'], C.fill_state_dict(r['keys']), 4), r['out'], **TOL)
    r = G['attn512']
    torch.testing.assert_close(OM.attention_block(r['x'], C.fill_state_dict(r['keys']), 4), r['out'], **TOL)
    r = G['resnet']
    torch.testing.assert_close(OM.resnet_block(r['x'], r['emb'], C.fill_state_dict(r['keys'])), r['out'], **TOL)
    r = G['lr']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(OU.lr_forward(sd, C.TINY_LR_CFG, r['x'], r['t'], r['x_self_cond']), r['out'], **TOL)
    torch.testing.assert_close(OU.lr_forward(sd, C.TINY_LR_CFG, r['x'], r['t']), r['out_nosc'], **TOL)
    oc, doc = small(G['split_small'])
    m = G['lr_mid']
    torch.testing.assert_close(OU.lr_forward_as_middle(sd, C.TINY_LR_CFG, m['h'], doc, m['t'], None),
                               m['out'], **TOL)


def test_dense_boundary_blocks(golden):
    """g_boundary: the reference's ConvDownsample / ConvUpsample / same-width ResnetBlock on dense tensors
    (modules.py:63-95, 474-513) -- the oracle's restatements of them (conv3d stride 2; nearest x2 + conv3d;
    resnet_block with an identity skip) against the reference's outputs."""
    import torch.nn.functional as F
    G = golden('g_boundary')
    r = G['convdown']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(F.conv3d(r['x'], sd['op.weight'], sd['op.bias'], stride=2, padding=1), r['out'], **TOL)
    r = G['convup']
    sd = C.fill_state_dict(r['keys'])
    torch.testing.assert_close(F.conv3d(F.interpolate(r['x'], scale_factor=2, mode='nearest'), sd['conv.weight'],
                                        sd['conv.bias'], padding=1), r['out'], **TOL)
    r = G['resnet_same']
    torch.testing.assert_close(OM.resnet_block(r['x'], r['emb'], C.fill_state_dict(r['keys'])), r['out'], **TOL)


def split_union_sd(sd):
    return {p: OM._sub(sd, p) for p in ('unet_lr', 'unet_hr', 'unet_feature')}


def test_unet(golden):
    G = golden('g_unet')
    oc, doc = small(G['split_small'])
    for name in ['uncond', 'cond']:
        r = G[name]
        parts = split_union_sd(C.fill_state_dict(r['keys']))
        hr = dict(C.TINY_HR_CFG, num_classes=r['num_classes'])
        lr = dict(C.TINY_LR_CFG, num_classes=r['num_classes'])
        y = OU.hr_forward(parts['unet_hr'], hr, r['x'], doc, r['t'], r['label'], parts['unet_lr'], lr)
        torch.testing.assert_close(y, r['out'], rtol=2e-4, atol=2e-5)
    r = G['feature']
    parts = split_union_sd(C.fill_state_dict(r['keys']))
    oc_l = OS.split2octree_large(oc, r['split_large'], 5)
    doc_l = OD.OracleDualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    cfg = r['cfg']
    feat = dict(input_depth=7, full_depth=3, model_channels=cfg['model_channels'][2],
                channel_mult=cfg['channel_mult'][2], num_res_blocks=cfg['num_res_blocks'][2], num_classes=None)
    mid = dict(kind='hr', input_depth=5, full_depth=3, model_channels=cfg['model_channels'][1],
               channel_mult=cfg['channel_mult'][1], num_res_blocks=cfg['num_res_blocks'][1], num_classes=None)
    y = OU.hr_forward(parts['unet_feature'], feat, r['x'], doc_l, r['t'], None, parts['unet_hr'], mid)
    torch.testing.assert_close(y, r['out'], rtol=2e-4, atol=2e-5)


def test_float64_oracle_is_the_same_algorithm(golden):
    """The precision contract (tests/test_gpu_precision.py, profiles/r03/precision_attribution.json) measures the product
    against the oracle run in float64 (oracle.modules.working_float).  That yardstick is pinned here, on the CPU: on
    the golden nets the float64 run must sit as close to the REFERENCE's recorded output as the float32 run does (same
    op sequence -- the difference between the two is rounding noise, far below the test tolerance of the goldens), and
    its result must really be float64 all the way (a silent down-cast anywhere would leave fp32-sized noise between two
    float64 runs at different thread counts / summation blockings; here: exact equality of two runs, and a distance to
    the float32 run that is of fp32 size, not zero)."""
    from oracle import modules as OM
    G = golden('g_unet')
    oc, doc = small(G['split_small'])

    def dbl(x):
        if isinstance(x, dict):
            return {k: dbl(v) for k, v in x.items()}
        return x.double() if torch.is_tensor(x) and x.is_floating_point() else x

    for name in ['uncond', 'cond']:
        r = G[name]
        parts = split_union_sd(C.fill_state_dict(r['keys']))
        hr = dict(C.TINY_HR_CFG, num_classes=r['num_classes'])
        lr = dict(C.TINY_LR_CFG, num_classes=r['num_classes'])
        y32 = OU.hr_forward(parts['unet_hr'], hr, r['x'], doc, r['t'], r['label'], parts['unet_lr'], lr)
        with OM.working_float(torch.float64):
            p64 = dbl(parts)
            y64 = OU.hr_forward(p64['unet_hr'], hr, r['x'].double(), doc, r['t'].double(), r['label'], p64['unet_lr'], lr)
            y64b = OU.hr_forward(p64['unet_hr'], hr, r['x'].double(), doc, r['t'].double(), r['label'], p64['unet_lr'], lr)
        assert OM.FLOAT == torch.float32                       # the context manager restores the reference arithmetic
        assert y64.dtype == torch.float64 and torch.equal(y64, y64b)
        scale = float(r['out'].abs().max())
        d_ref = float((y64.float() - r['out']).abs().max()) / scale
        d_32 = float((y32 - r['out']).abs().max()) / scale
        d_6432 = float((y64 - y32.double()).abs().max()) / scale
        assert d_ref < 2e-5 and d_32 < 2e-5, (name, d_ref, d_32)
        assert 1e-9 < d_6432 < 2e-5, (name, d_6432)            # fp32-sized, and not zero: the two runs do differ in arithmetic


def fake_net(shape):
    A = torch.linspace(-0.5, 0.5, shape[1])

    def net(x, noise_cond, x_self_cond):
        a = A.view(1, -1, *([1] * (x.ndim - 2)))
        tt = noise_cond.view(-1, *([1] * (x.ndim - 1))) if x.ndim > 2 else noise_cond[0]
        y = torch.tanh(x * 0.7 + a) * 0.9 + 0.05 * torch.tanh(tt)
        if x_self_cond is not None:
            y = y + 0.1 * x_self_cond
        return y
    return net


@pytest.mark.parametrize('name,unet_type,df_type', [('x0', 'lr', 'x0'), ('eps', 'hr', 'eps'), ('x0_graph', 'hr', 'x0')])
def test_sample_loop(golden, name, unet_type, df_type):
    r = golden('g_sample_loop')[name]
    torch.manual_seed(r['seed'])
    y = OS.sample_loop(fake_net(r['shape']), r['shape'], r['B'], r['steps'], unet_type, df_type,
                       r.get('trunc', 0.0))
    torch.testing.assert_close(y, r['out'], **TOL)


def test_vae_decoder(golden):
    from oracle import vae as OV
    G = golden('g_vae')
    oc, doc = tiny(G['split_small'])
    oc_l = OS.split2octree_large(oc, G['split_large'], 4)
    doc_l = OD.OracleDualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    sd = C.fill_state_dict(G['keys'])
    code = C.rand_input('vae_code', doc_l.graph[4]['node_type'].numel(), 3)
    logits, regs, _ = OV.decode_code(sd, G['cfg'], code, doc_l, update_octree=False)
    for d in (4, 5, 6):
        torch.testing.assert_close(logits[d], G['logits'][d], rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(regs[d], G['reg_voxs'][d], rtol=2e-4, atol=2e-5)
    # growth path: same argmax decisions -> same octree (the logits at every depth are pinned too)
    import copy
    doc_in = OD.OracleDualOctree(copy.deepcopy(oc_l))
    doc_in.post_processing_for_docnn()
    logits2, regs2, oct2 = OV.decode_code(sd, G['cfg'], code, doc_in, update_octree=True)
    assert torch.equal(oct2.nnum, G['grow']['nnum']) and torch.equal(oct2.nnum_nempty, G['grow']['nnum_nempty'])
    for d in (4, 5, 6):
        torch.testing.assert_close(logits2[d], G['grow']['logits'][d], rtol=2e-4, atol=2e-5)
        assert tuple(regs2[d].shape) == G['grow']['reg_shapes'][d]


def mpu_case(G):
    oc, _ = tiny(G['split_small'])
    oc_l = OS.split2octree_large(oc, G['split_large'], 4)
    fd, ds, dp = G['cfg']
    ncum = torch.cumsum(oc_l.nnum, 0)
    reg = {d: C.rand_input('mpu_code_%d' % d, int(ncum[d] - (ncum[fd - 1] if fd else 0)), 4)
           for d in range(ds, dp + 1)}
    return oc_l, reg, (fd, ds, dp)


def test_neural_mpu(golden):
    """oracle/mpu.py against the reference's own mpu.py outputs (tests/golden/g_mpu.pt)."""
    from oracle import mpu as OMPU
    G = golden('g_mpu')
    oc_l, reg, (fd, ds, dp) = mpu_case(G)
    out = OMPU.neural_mpu(G['pos'], reg, oc_l, fd, ds, dp)
    for d in range(ds, dp + 1):
        assert torch.equal(out[d][1], G['mask'][d])
        torch.testing.assert_close(out[d][0], G['sdf'][d], rtol=1e-4, atol=1e-5)


def test_split_roundtrip_oracle():
    """octree -> (split_small, split_large) -> octree is the identity on keys (util_dualoctree.py:199-273)."""
    oc6 = OS.split2octree_small(C.shell6_split(2, jitter=True), 6, 4)
    x, y, z, b = oc6.xyzb(6)
    sl = C.random_split_large(int(oc6.nnum[6]), 3, p=0.4)
    oc8 = OS.split2octree_large(oc6, sl, 6)
    ss = OS.octree2split_small(oc8, 4)
    assert ss.shape == (2, 8, 16, 16, 16) and set(ss.unique().tolist()) <= {-1.0, 1.0}
    back6 = OS.split2octree_small(ss, 6, 4)
    back8 = OS.split2octree_large(back6, OS.octree2split_large(oc8, 6), 6)
    for d in range(9):
        assert torch.equal(back8.keys[d], oc8.keys[d])
        if d < 8:
            assert torch.equal(back8.children[d], oc8.children[d])
    # split_large of a grown tree marks exactly the children that were grown
    assert torch.equal(OS.octree2split_large(oc8, 6) > 0, (sl > 0) & ((sl > 0).any(1, keepdim=True)))


def test_vae_encoder(golden):
    from oracle import vae as OV
    G = golden('g_vae_enc')
    oc, _ = tiny(G['split_small'])
    oc_l = OS.split2octree_large(oc, G['split_large'], 4)
    doc_l = OD.OracleDualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    sd = C.fill_state_dict(G['keys'])
    data = C.rand_input('vae_enc_in', doc_l.graph[6]['node_type'].numel(), 4)
    h, kl = OV.encode(sd, G['cfg'], data, doc_l)
    torch.testing.assert_close(h[::8], G['h_rows8'], rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(kl, G['kl'], rtol=2e-4, atol=2e-5)


def sub_close(actual, rec, rtol=2e-3, atol_rel=2e-4):
    """compare a tensor with a make_golden `sub` record (strided sample + L2 norm)."""
    f = actual.detach().reshape(-1)
    assert tuple(actual.shape) == tuple(rec['shape'])
    scale = float(rec['vals'].abs().max()) + 1e-30
    torch.testing.assert_close(f[::rec['stride']], rec['vals'], rtol=rtol, atol=atol_rel * scale)
    assert abs(float(f.double().norm()) - rec['norm']) <= 2e-3 * rec['norm'] + 1e-12


def vae_train_case(G):
    oc, _ = tiny(G['split_small'])
    oc_l = OS.split2octree_large(oc, G['split_large'], 4)
    doc_l = OD.OracleDualOctree(oc_l)
    doc_l.post_processing_for_docnn()
    sd = C.fill_state_dict(G['keys'])
    data = C.rand_input('vae_enc_in', doc_l.graph[6]['node_type'].numel(), 4)
    noise = C.rand_input('vae_post_noise', *G['n_noise'])
    return oc_l, doc_l, sd, data, noise


def test_vae_training_step(golden):
    """oracle forward_train + geometry_loss + autograd against the reference's own GraphVAE.forward /
    loss.geometry_loss / backward (tests/golden/g_vae_train.pt): every named loss, the MPU gradients, the
    gradients w.r.t. logits / reg_voxs and w.r.t. every parameter."""
    from oracle import loss as OL
    from oracle import vae as OV
    G = golden('g_vae_train')
    oc_l, doc_l, sd, data, noise = vae_train_case(G)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pos = G['pos'].clone().requires_grad_(True)
    with torch.enable_grad():
        out = OV.forward_train(sd, G['cfg'], data, doc_l, doc_l, pos, noise)
        for t in list(out['logits'].values()) + list(out['reg_voxs'].values()):
            t.retain_grad()
        mg = OL.mpu_gradients(out['mpus'], pos)
        losses = OL.geometry_loss(out['logits'], out['mpus'], oc_l, pos, G['sdf_gt'], G['grad_gt'], out['kl'],
                                  G['kl_weight'])
        total = OL.total_loss(losses)
        total.backward()
    for k, v in G['losses'].items():
        assert abs(float(losses[k]) - v) <= 2e-4 * abs(v) + 1e-6, (k, float(losses[k]), v)
    assert abs(float(total) - G['total']) <= 2e-4 * G['total']
    for d in (4, 5, 6):
        sub_close(out['logits'][d], G['logits'][d])
        sub_close(out['reg_voxs'][d], G['reg_voxs'][d])
        torch.testing.assert_close(out['mpus'][d][0].detach(), G['sdf'][d], rtol=1e-3, atol=1e-4)
        g_ref = G['mpu_grad'][d]
        torch.testing.assert_close(mg[d].detach(), g_ref, rtol=2e-3, atol=2e-4 * float(g_ref.abs().max()))
        sub_close(out['logits'][d].grad, G['d_logits'][d])
        sub_close(out['reg_voxs'][d].grad, G['d_reg_voxs'][d])
    n = 0
    for k, rec in G['grads'].items():
        if rec is None:
            assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0, k
            continue
        sub_close(sd[k].grad, rec, rtol=5e-3, atol_rel=1e-3)
        n += 1
    assert n > 100


def test_points_octree_oracle():
    """oracle/points.py (ocnn's bottom-up build, restated) against a brute-force occupancy pyramid, and through the
    reference's own octree <-> split codes round trip (utils/util_dualoctree.py:199-273)."""
    from oracle import points as OP
    depth, fd = 6, 4
    clouds = [C.surface_points(3000, 7, 'sphere'), C.surface_points(2500, 8, 'torus')]
    oc, feat = OP.points2octree_batch([c[0] for c in clouds], [c[1] for c in clouds], depth, fd)
    assert oc.batch_size == 2 and feat.shape == (int(oc.nnum[depth]), 4)
    for b, (pts, _) in enumerate(clouds):
        ijk = ((pts + 1.0) * 2 ** (depth - 1)).long().clamp(0, 2 ** depth - 1)
        for d in range(fd, depth + 1):
            occ = torch.zeros(2 ** d, 2 ** d, 2 ** d, dtype=torch.bool)
            c = ijk >> (depth - d)
            occ[c[:, 0], c[:, 1], c[:, 2]] = True
            x, y, z, bb = oc.xyzb(d)
            sel = bb == b
            assert torch.equal(oc.nempty_mask(d)[sel], occ[x[sel], y[sel], z[sel]])      # non-empty <=> holds a point
            assert int(occ.sum()) == int(oc.nempty_mask(d)[sel].sum())                   # and no occupied cell is missing
    for d in range(depth + 1):
        assert bool((torch.diff(oc.keys[d]) > 0).all())                                  # sorted, batch-major
        ne = oc.children[d][oc.children[d] >= 0]
        assert torch.equal(ne, torch.arange(ne.numel(), dtype=torch.int32))
        if d < depth:
            assert int(oc.nnum[d + 1]) == 8 * int(oc.nnum_nempty[d])
    # the reference's data formats round-trip the structure
    back = OS.split2octree_small(OS.octree2split_small(oc, fd), depth, fd)
    for d in range(depth + 1):
        assert torch.equal(back.keys[d], oc.keys[d])
        if d < depth:
            assert torch.equal(back.children[d], oc.children[d])
    # feature: unit normals on non-empty nodes, |D| <= sqrt(3)/2, zero rows elsewhere
    ne = oc.nempty_mask(depth)
    torch.testing.assert_close(feat[ne, :3].norm(dim=1), torch.ones(int(ne.sum())), rtol=1e-5, atol=1e-5)
    assert float(feat[~ne].abs().max()) == 0.0 and float(feat[:, 3].abs().max()) <= 0.867


def test_sketch_detects_a_localised_error():
    """tests/golden/oracle_cache.py: the block projections of a Sketch see ONE wrong row and one wrong 64 x 32 tile
    anywhere in a tensor whose row sample holds 0.3 % of the rows ('a handful of wrong rows at a
    tile / share / XCD boundary passes'), at the tolerances the GPU tests use; rounding-level noise stays below them."""
    import oracle_cache as OC
    g = torch.Generator().manual_seed(5)
    t = torch.randn(100_000, 128, generator=g)
    sk = OC.unpack(OC.pack('unit', t))
    assert isinstance(sk, OC.Sketch) and sk.idx.numel() < 400 and tuple(sk.proj.shape) == ((100_000 + 63) // 64, OC.NPROJ)
    scale = float(t.abs().max())
    y = t + 1e-6 * scale * torch.randn(t.shape, generator=g)           # fp16x3-level rounding noise
    e0 = OC.errors(y, sk)
    assert e0['rel_to_max'] < 1e-5 and e0['rows_covered'] == 100_000, e0
    row = 51_234
    assert row not in set(sk.idx.tolist())
    y1 = y.clone()
    y1[row] += 2e-4 * scale                                             # one row, every element off by 2e-4 of the range
    e1 = OC.errors(y1, sk)
    assert e1['proj_rel'] > 2e-5 and e1['rel_to_max'] > 2e-5, e1        # (the per-layer sweep's bar is 2e-5)
    y2 = y.clone()
    y2[64 * 700:64 * 701, 32:64] += 1e-4 * scale                        # one 64-row x 32-column sub-tile
    e2 = OC.errors(y2, sk)
    assert e2['rel_to_max'] > 2e-5, e2
    y3 = y.clone()
    y3[99_990, 5] = 0.0                                                 # one element lost (e.g. a clamped column)
    assert OC.errors(y3, sk)['rel_to_max'] > 1e-4 or abs(float(t[99_990, 5])) < 1e-3 * scale
    # Full / Rows wrappers
    f = OC.unpack(OC.pack('unit_full', OC.Full(t)))
    assert torch.is_tensor(f) and torch.equal(f, t)
    sr = OC.unpack(OC.pack('unit_rows', OC.Rows(t, torch.tensor([row, 70_000]))))
    assert {row, 70_000} <= set(sr.idx.tolist())


def test_oracle_cache_fixtures_are_current():
    """Every committed tests/golden/oracle_cache/*.pt carries the digest of the tree: oracle/*.py, the input builders,
    oracle_cache.py AND the test module that registers the case (an edited oracle invocation must invalidate
    its fixture).  The GPU tests assert the same when they read a fixture; this is the CPU-side check of all of them."""
    import importlib
    import os
    import oracle_cache as OC
    import make_oracle_cache as MK
    for m in MK.MODULES:
        importlib.import_module(m)
    files = sorted(f for f in os.listdir(OC.DIR) if f.endswith('.pt'))
    assert len(files) >= 50
    for f in files:
        name = f[:-3]
        assert name in OC.REGISTRY, 'fixture %s has no registered case' % f
        rec = torch.load(os.path.join(OC.DIR, f), weights_only=False)
        assert rec['digest'] == OC.case_digest(name), (f, rec['digest'], OC.case_digest(name))
    assert set(OC.REGISTRY) == {f[:-3] for f in files}, 'cases without a fixture: %s' % sorted(set(OC.REGISTRY) - {f[:-3] for f in files})
