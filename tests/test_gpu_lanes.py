"""A graph stage as lanes: runs of consecutive shapes of the batch, each on its own HIP stream (octfusion_amd/sampler.py).

The reference runs the whole batch through one forward (models/octfusion_model_union.py:293-352); nothing in the network
couples the shapes of a batch (dual_octree.py: graphs never cross batch elements; modules.py:291-326: GroupNorm statistics
per element), so the lanes must reproduce it:

  * Octree.batch_slices / DualOctree.split_batch against octrees BUILT from the slice of the split codes: same keys,
    child pointers, node counts and CSR graphs, and a row map that picks exactly the part's rows in the part's order
    (ragged batch with an empty element, uneven part sizes, shell-8 trees grown by split2octree_large);
  * sample_loop with 2 and 3 lanes against the one-lane call (eager and hipGraph replay, with labels, with explicit
    per-step noise on the x0 branch): equal to fp32 rounding of the summation order inside a launch;
  * the persistent launch planned for fewer compute units (ofx_set_gconv_cus, what a lane uses) against the default plan.
"""
import pytest
import torch

import common as C
from test_gpu_fullwidth import dev, errors

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def _octree(split, large=False):
    from octfusion_amd import synthetic
    from octfusion_amd.octree import split2octree_large, split2octree_small
    S = split.shape[-1]
    fd = {8: 3, 16: 4}[S]
    oc = split2octree_small(split.to(dev()), fd + 2, fd)
    if large:
        x6, y6, z6, _ = oc.xyzb(fd + 2)
        oc = split2octree_large(oc, synthetic.shell8_split_large(x6, y6, z6), fd + 2)
    return oc


@pytest.mark.parametrize('kind,parts', [('ragged', 2), ('ragged', 3), ('shell', 2), ('shell8', 2)])
def test_split_batch_equals_octrees_built_from_the_slices(kind, parts):
    from octfusion_amd import synthetic
    from octfusion_amd.dual_octree import DualOctree
    if kind == 'ragged':
        split = C.random_split_small(5, 3, 23, p=0.45)
        split[1] = -1.0                                   # an element with nothing below the full layer
    else:
        split = C.shell6_split(4 if kind == 'shell' else 2, jitter=True)
    large = kind == 'shell8'
    oc = _octree(split, large)
    for d in range(oc.depth + 1):
        k = oc.keys[d]
        assert bool((k[1:] > k[:-1]).all()), 'keys of depth %d are not strictly ascending' % d
    doc = DualOctree(oc)
    B = split.shape[0]
    got = doc.split_batch(parts)
    assert [p[2] for p in got] == [(B * i // parts, B * (i + 1) // parts) for i in range(parts)]
    seen = torch.zeros(doc.total_num, dtype=torch.int32, device=dev())
    for sub, rows, (b0, b1) in got:
        if large:
            # the large tree of a slice: grow the slice's small tree with the slice of the depth-6 split codes
            oc_s = _octree(split[b0:b1], False)
            x6, y6, z6, b6 = oc.xyzb(oc.depth - 2)
            sl = synthetic.shell8_split_large(x6, y6, z6)
            from octfusion_amd.octree import split2octree_large
            oc_w = split2octree_large(oc_s, sl[(b6 >= b0) & (b6 < b1)], oc_s.depth)
        else:
            oc_w = _octree(split[b0:b1])
        want = DualOctree(oc_w)
        assert sub.batch_size == b1 - b0 and sub.depth == want.depth
        assert torch.equal(sub.nnum, want.nnum) and torch.equal(sub.nenum, want.nenum)
        for d in range(oc.depth + 1):
            assert torch.equal(sub.octree.keys[d], want.octree.keys[d]), d
            assert torch.equal(sub.octree.children[d], want.octree.children[d]), d
        for d in range(sub.full_depth, sub.depth + 1):
            for a, b in zip(sub.csr(d)[:2], want.csr(d)[:2]):
                assert torch.equal(a, b)
            assert torch.equal(sub.batch_id(d), want.batch_id(d))
        bid = doc.batch_id(doc.depth)[rows]
        assert torch.equal(bid - b0, sub.batch_id(sub.depth).to(bid.dtype))
        assert bool((rows[1:] > rows[:-1]).all())
        seen[rows] += 1
    assert bool((seen == 1).all())


def _net(cond):
    from octfusion_amd import configs, graph_unet_union as U, synthetic
    cfg = dict(configs.SNET_COND if cond else configs.SNET_UNCOND, model_channels=[64, 64])
    net = U.UNet3DModel(**{k: v for k, v in dict(cfg, stage_flag='hr').items() if k != 'df_type'})
    net.load_state_dict(synthetic.random_state_dict(net))
    return net.to(dev()).eval()


@pytest.mark.parametrize('cond,df,lanes,use_graph', [(False, 'eps', 2, True), (False, 'eps', 3, False), (True, 'eps', 2, True),
                                                       (False, 'x0', 2, True)])
def test_lanes_reproduce_the_one_lane_call(cond, df, lanes, use_graph):
    from octfusion_amd import ops, sampler
    from octfusion_amd.dual_octree import DualOctree
    split = C.shell6_split(4, jitter=True)
    doc = DualOctree(_octree(split))
    net = _net(cond)
    shp = (doc.total_num, 3)
    g = torch.Generator().manual_seed(11)
    init = torch.randn(shp, generator=g)
    steps = 3
    noise = [torch.randn(shp, generator=g) for _ in range(steps)] if df == 'x0' else None
    label = (torch.arange(4) % 5).to(dev()) if cond else None
    kw = dict(doctree=doc, unet_lr=net.unet_lr, label=label, init_noise=init, step_noise=noise)
    one = sampler.sample_loop(net, shp, 4, steps, 'hr', df, dev(), use_graph=use_graph, lanes=1, **kw)
    many = sampler.sample_loop(net, shp, 4, steps, 'hr', df, dev(), use_graph=use_graph, lanes=lanes, **kw)
    torch.cuda.synchronize()
    assert not ops.sync_error(dev())
    e = errors(many, one.double())
    # one step differs by ~2e-6 of the largest value (k-split of the persistent launch follows the row count; fp64 atomics of
    # the fused statistics); a random-weight net amplifies that by up to ~3x per step
    assert e['rel_to_max'] < 5e-5, e
    # and the default picks lanes for this call (graph stage, batch >= 2, replayed) without being asked
    assert sampler.lane_count(4, doc, True) == min(sampler.LANES, 4) and sampler.lane_count(4, doc, False) == 1
    assert sampler.lane_count(1, doc, True) == 1 and sampler.lane_count(4, None, True) == 1


def test_a_part_without_nodes_falls_back_to_one_lane():
    """Three elements of which the last has nothing below the full layer: with three lanes its part has no nodes at
    depth full_depth + 1, and the call runs on one stream instead."""
    from octfusion_amd import sampler
    from octfusion_amd.dual_octree import DualOctree
    split = C.shell6_split(3, jitter=True)
    split[2] = -1.0
    doc = DualOctree(_octree(split))
    net = _net(False)
    shp = (doc.total_num, 3)
    init = torch.randn(shp, generator=torch.Generator().manual_seed(5))
    kw = dict(doctree=doc, unet_lr=net.unet_lr, init_noise=init, use_graph=False)
    a = sampler.sample_loop(net, shp, 3, 2, 'hr', 'eps', dev(), lanes=1, **kw)
    b = sampler.sample_loop(net, shp, 3, 2, 'hr', 'eps', dev(), lanes=3, **kw)
    assert torch.equal(a, b)


def test_graphconv_planned_for_fewer_compute_units():
    """ofx_set_gconv_cus(n): same products, another cut of the stream-K region -> fp32 rounding of the piece order."""
    from octfusion_amd import _lib, modules as M, ops
    from octfusion_amd.dual_octree import DualOctree
    doc = DualOctree(_octree(C.shell6_split(2, jitter=True)))
    d, cin, cout = 6, 128, 128
    conv = M.GraphConv(cin, cout, 7, 7, 7, use_bias=True)
    conv.load_state_dict(C.fill_state_dict([(k, tuple(v.shape)) for k, v in conv.state_dict().items()]))
    conv = conv.to(dev())
    x = C.rand_input('lanes_cus', doc.csr(d)[2], cin).to(dev())
    base = conv(x, doc, d)
    try:
        for cus in (160, 96, 8):
            ops.set_lane_cus(cus)
            y = conv(x, doc, d)
            e = errors(y, base.double())
            assert e['rel_to_max'] < 2e-6, (cus, e)
        with pytest.raises(_lib.OfxError):
            ops.set_lane_cus(3)
        ops.set_lane_cus(100000)                 # above the device's count: clamped to it -> the default plan, same bits
        assert torch.equal(conv(x, doc, d), base)
    finally:
        ops.set_lane_cus(0)
    assert torch.equal(conv(x, doc, d), base)
    assert not ops.sync_error(dev())


def test_lanes_at_the_bench_size():
    """configs[2] at its real widths, eight shapes (N6 = 217 008), two lanes of four on two streams with the persistent
    launches planned for LANE_CUS compute units, hipGraph replay: the whole stage against the one-lane call."""
    from octfusion_amd import configs, ops, sampler, synthetic
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.graph_unet_union import UNet3DModel
    net = UNet3DModel(**configs.unet_params('snet_uncond', 'hr'))
    net.load_state_dict(synthetic.random_state_dict(net))
    net = net.to(dev()).eval()
    doc = DualOctree(_octree(synthetic.shell6_split(8, jitter=True)))
    shp = (doc.total_num, 3)
    init = torch.randn(shp, generator=torch.Generator().manual_seed(3))
    kw = dict(doctree=doc, unet_lr=net.unet_lr, init_noise=init, use_graph=True)
    one = sampler.sample_loop(net, shp, 8, 3, 'hr', 'eps', dev(), lanes=1, **kw)
    two = sampler.sample_loop(net, shp, 8, 3, 'hr', 'eps', dev(), lanes=2, **kw)
    torch.cuda.synchronize()
    assert not ops.sync_error(dev())
    e = errors(two, one.double())
    assert e['rel_to_max'] < 5e-5, e
