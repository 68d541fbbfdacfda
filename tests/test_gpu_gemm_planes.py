"""The unpool GEMM on the data path of the planes GraphConv (`ofx_gemm_planes`, csrc/ofx_gemm3.hip with one direction;
reference models/networks/modules.py:430-446 `Upsample`: x[n, C] @ W.flatten(1) -> [n, 8 C], call site :458-467).

  * against x.double() @ W.double() for every shape class: whole-tile rounds + a stream-K region (M large), fewer tiles
    than blocks (cut tiles: pieces handed over through the workspace), 8 and 16 k-steps per tile, both pair modes, fp32 and
    pair-plane outputs, a bias, M not a multiple of the row tile, N not a multiple of the column tile;
  * shapes that do not qualify return "not launched" and the caller's register-staged kernel answers;
  * `unpool_nodes` (GraphUpsample's first half) with the switch on and off on the ragged tree: same rows to 2e-6.
"""
import pytest
import torch

import common as C
from test_gpu_fullwidth import dev, errors

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


@pytest.mark.parametrize('M,K,N,mode,out_mode,bias', [
    (21344, 256, 2048, 3, 3, False), (4976, 512, 4096, 3, 3, False), (1000, 256, 2048, 3, 0, True),
    (333, 512, 384, 2, 2, False), (70001, 256, 256, 3, 0, False), (5000, 768, 1024, 3, 3, True)])
def test_gemm_planes_vs_float64(M, K, N, mode, out_mode, bias):
    from octfusion_amd import ops
    saved = ops.get_precision()
    ops.set_precision('fp16x3' if mode == 3 else 'bf16x3')
    try:
        x = C.rand_input('gp_x_%d_%d' % (M, K), M, K)
        w = C.rand_input('gp_w_%d_%d' % (K, N), K, N) * (1.5 / K ** 0.5)
        b = C.rand_input('gp_b_%d' % N, N) if bias else None
        ref = x.double() @ w.double() + (b.double() if bias else 0.0)
        xp = ops.planes_split(x.to(dev()), mode)
        wg = w.to(dev())
        pgp = ops.PackedGemmPlanes().get(wg, mode)
        out = torch.full((M, N), float('nan'), device=dev())
        ok = ops.gemm_planes(xp, pgp, out, out_mode, bias=b.to(dev()) if bias else None)
        assert ok, 'shape did not qualify'
        if out_mode:
            setattr(out, ops.PLANES_ATTR, out_mode)
            y = ops.planes_merge(out, out_mode)
        else:
            y = out
        e = errors(y, ref)
        assert e['rel_to_max'] < (2e-6 if mode == 3 else 2e-5), e
        assert not ops.sync_error(dev())
        # a transposed weight view (strides (1, K)) packs to the same planes
        pgp2 = ops.PackedGemmPlanes().get(wg.t().contiguous().t(), mode)
        nb = (K // 32) * N * 128 + 8                 # planes + the two scale words of the trailer (the rest of it is unused)
        assert torch.equal(pgp.t[:nb], pgp2.t[:nb])
    finally:
        ops.set_precision(saved)


def test_gemm_planes_declines_shapes_it_is_not_built_for():
    from octfusion_amd import ops
    for M, K, N in ((5000, 128, 1024), (4000, 256, 64), (40, 256, 256)):       # 4 k-steps; narrow N; a handful of units
        xp = ops.planes_split(torch.randn(M, K, device=dev()), 3)
        pgp = ops.PackedGemmPlanes().get(torch.randn(K, N, device=dev()), 3)
        out = torch.zeros(M, N, device=dev())
        assert not ops.gemm_planes(xp, pgp, out, 0)
        assert float(out.abs().max()) == 0.0                                   # nothing was launched


def test_unpool_nodes_through_the_planes_gemm():
    from octfusion_amd import modules as M, ops
    from octfusion_amd.dual_octree import DualOctree
    from octfusion_amd.octree import split2octree_small
    split = C.shell6_split(2, jitter=True)
    doc = DualOctree(split2octree_small(split.to(dev()), 6, 4))
    d, Cc = 5, 256
    up = M.Upsample(Cc)
    up.load_state_dict(C.fill_state_dict([(k, tuple(v.shape)) for k, v in up.state_dict().items()]))
    up = up.to(dev())
    x = C.rand_input('gp_unpool', doc.csr(d)[2], Cc).to(dev())
    mode = ops.planes_mode()
    saved = ops.GEMM_PLANES
    outs = []
    try:
        for on in (True, False):
            ops.GEMM_PLANES = on
            y = M.unpool_nodes(x, doc, d, up, planes=mode)
            assert ops.planes_of(y) == mode
            outs.append(ops.planes_merge(y, mode))
    finally:
        ops.GEMM_PLANES = saved
    copy_src, a_rows, n_copy = doc.unpool_maps(d)
    assert a_rows.numel() >= 256
    ref = torch.cat([x[copy_src.long()].double(), (x[a_rows.long()].double() @ up.weights.double().view(Cc, 8 * Cc)).view(-1, Cc)])
    for y in outs:
        assert errors(y, ref)['rel_to_max'] < 2e-6
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-6 * float(ref.abs().max())
