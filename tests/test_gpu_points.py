"""GPU parity of the point cloud -> octree build: the batched top-down build on libofx against the
oracle's restatement of ocnn's bottom-up build + merge (oracle/points.py; ocnn itself is absent: parity at that
boundary is unpinned, see the oracle header).  Keys / child pointers / counts bit-exact; 'ND' feature to 1e-5."""
import pytest
import torch

import common as C
from test_gpu_parity import close, dev

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def clouds(sizes_kinds, seed0=7):
    return [C.surface_points(n, seed0 + i, k) for i, (n, k) in enumerate(sizes_kinds)]


def to_points(cl):
    from octfusion_amd.octree import Points
    return [Points(p.to(dev()), n.to(dev())) for p, n in cl]


def same_octree(oc, ref):
    assert oc.depth == ref.depth and oc.batch_size == ref.batch_size
    for d in range(ref.depth + 1):
        assert int(oc.nnum[d]) == int(ref.nnum[d]) and int(oc.nnum_nempty[d]) == int(ref.nnum_nempty[d]), d
        assert torch.equal(oc.keys[d].cpu(), ref.keys[d]), d
        assert torch.equal(oc.children[d].cpu(), ref.children[d]), d


@pytest.mark.parametrize('depth,fd', [(6, 4), (8, 4), (6, 2)])
def test_build_octree_batch_vs_oracle(depth, fd):
    from oracle import points as OP
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import build_octree_batch, merge_octrees, Octree
    cl = clouds([(20000, 'sphere'), (15000, 'torus'), (300, 'sphere')])
    ref, feat_ref = OP.points2octree_batch([c[0] for c in cl], [c[1] for c in cl], depth, fd)
    oc = build_octree_batch(to_points(cl), depth, fd)
    same_octree(oc, ref)
    feat = oc.get_input_feature('ND')
    close(feat, feat_ref, 1e-5)
    assert float(feat[~oc.nempty_mask(depth)].abs().max()) == 0.0
    # per-shape builds + merge_octrees (the reference's call pattern) give the same octree and feature
    singles = [Octree(depth, fd, 1, dev()).build_octree(p) for p in to_points(cl)]
    merged = merge_octrees(singles)
    same_octree(merged, ref)
    assert torch.equal(merged.get_input_feature('ND'), feat)
    # the dual graph builds on it and the encoder input has one row per graph node
    doc = DualOctree(oc)
    data = doc.get_input_feature()
    assert data.shape == (doc.csr(depth)[2], 4)
    assert torch.equal(data[data.shape[0] - feat.shape[0]:], feat) and float(data[:data.shape[0] - feat.shape[0]].abs().max()) == 0.0


def test_build_octree_roundtrip_and_edges():
    """structure -> split codes -> structure (utils/util_dualoctree.py:199-273) is the identity; degenerate clouds
    (one point, all points in one cell, an element with a single point next to a dense one) build correctly."""
    from oracle import points as OP
    from octfusion_amd.octree import (build_octree_batch, octree2split_large, octree2split_small, split2octree_large,
                                      split2octree_small, Points)
    cl = clouds([(30000, 'torus'), (8000, 'sphere')], seed0=21)
    oc = build_octree_batch(to_points(cl), 8, 4)
    back6 = split2octree_small(octree2split_small(oc, 4), 6, 4)
    back8 = split2octree_large(back6, octree2split_large(oc, 6), 6)
    for d in range(9):
        assert torch.equal(back8.keys[d], oc.keys[d])
        if d < 8:
            assert torch.equal(back8.children[d], oc.children[d])
    one = (torch.tensor([[0.3, -0.2, 0.9]]), torch.tensor([[0.0, 0.0, 1.0]]))
    blob = (torch.full((64, 3), 0.123) + torch.rand(64, 3, generator=torch.Generator().manual_seed(1)) * 1e-4,
            torch.nn.functional.normalize(torch.randn(64, 3, generator=torch.Generator().manual_seed(2)), dim=1))
    for cl2 in ([one], [blob], [one, cl[1], blob]):
        ref, feat_ref = OP.points2octree_batch([c[0] for c in cl2], [c[1] for c in cl2], 6, 4)
        oc2 = build_octree_batch(to_points(cl2), 6, 4)
        same_octree(oc2, ref)
        close(oc2.get_input_feature('ND'), feat_ref, 1e-5)
    # Points.clip drops what lies outside the cube before the build (datasets/dualoctree_snet.py:45)
    # -- strictly: a point ON a face of the cube goes too (ocnn's inbox_mask is points > min and points < max)
    p = Points(torch.tensor([[0.0, 0.0, 0.0], [1.5, 0.0, 0.0], [0.2, -1.01, 0.1], [1.0, 0.0, 0.0], [0.3, -1.0, 0.2],
                             [0.999999, -0.999999, 0.5]]).to(dev()),
               torch.tensor([[1.0, 0.0, 0.0]] * 6).to(dev()))
    mask = p.clip(-1.0, 1.0)
    assert mask.tolist() == [True, False, False, False, False, True]
    assert p.points.shape[0] == 2 and p.normals.shape[0] == 2


def test_vae_encoder_runs_on_built_octree():
    """points -> octree -> dual graph -> 'ND' feature -> GraphVAE.encode -> decode_code end to end on device."""
    from octfusion_amd import synthetic
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.graph_vae import GraphVAE
    from octfusion_amd.octree import build_octree_batch
    cl = clouds([(40000, 'sphere'), (40000, 'torus')], seed0=3)
    oc = build_octree_batch(to_points(cl), 8, 4)
    doc = DualOctree(oc)
    vae = GraphVAE(depth=8, channel_in=4, nout=4, full_depth=4, depth_stop=6, depth_out=8, resblk_type='basic',
                   resblk_num=2, embed_dim=3)
    vae.load_state_dict(synthetic.random_state_dict(vae))
    vae = vae.to(dev()).eval()
    code, mean, logvar = vae.encode(doc.get_input_feature(), doc, sample=False)
    assert code.shape == (doc.csr(6)[2], 3) and bool(torch.isfinite(code).all())
    out = vae.decode_code(code, doc, update_octree=True)
    assert out['octree_out'].depth == 8 and all(bool(torch.isfinite(v).all()) for v in out['reg_voxs'].values())
