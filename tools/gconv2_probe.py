"""A/B probe of the two fused GraphConv kernels on the bench shapes (run on the GPU box):
register-staged 128 x 128 kernel (ofx_gemm.hip) vs LDS-DMA planes kernel (ofx_gemm2.hip, both scheduling
variants, bf16x3 and fp16), correctness against the exact-fp32 MFMA kernel and time per launch.

    python tools/gconv2_probe.py [--batch 8] [--quick]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops, synthetic, modules as M
from octfusion_amd.dual_octree import DualOctree
from octfusion_amd.octree import split2octree_small

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--quick', action='store_true')
ap.add_argument('--json', default=None)
args = ap.parse_args()

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
oc = split2octree_small(synthetic.shell6_split(args.batch).to(dev), 6, 4)
doc = DualOctree(oc)
B = args.batch


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max())


shapes = [(6, 128, 128), (6, 256, 128), (6, 384, 128), (5, 256, 256), (5, 512, 256), (6, 128, 256)]
if args.quick:
    shapes = shapes[:2]
rows = []
for d, cin, cout in shapes:
    N = doc.csr(d)[2]
    conv = M.GraphConv(cin, cout, 7, 7, d - 1).to(dev)
    conv.emit_stats = False
    x = torch.randn(N, cin, device=dev)
    emb = torch.randn(B, cout, device=dev)
    res = torch.randn(N, cout, device=dev)
    flops = 2.0 * N * 7 * (cin + d - 1) * cout

    ops.USE_PLANES = False
    ops.set_precision('fp32')
    y_ref = conv(x, doc, d, emb=emb, res=res).clone()
    ops.set_precision('bf16x3')
    y1 = conv(x, doc, d, emb=emb, res=res)
    e1 = relmax(y1, y_ref)
    t1 = timeit(lambda: conv(x, doc, d, emb=emb, res=res))

    ops.USE_PLANES = True
    ops.PLANES_MIN_TILES = 1
    out = dict(d=d, N=N, cin=cin, cout=cout, v1_ms=t1, v1_err=e1, v1_TF=flops / t1 / 1e9)
    for prec, mode in (('bf16x3', 2), ('fp16', 1)):
        if mode == 1 and cin % 64:
            continue
        ops.set_precision(prec)
        xp = ops.planes_split(x, mode)
        back = ops.planes_merge(xp, mode)
        out['%s_split_err' % prec] = relmax(back, x)
        for variant in (0, 1):          # here: tile geometry 4 (256 rows, 1 block/CU) vs 2 (128 rows, 2 blocks/CU)
            _lib.call('ofx_set_gconv2_tile', 4 if variant == 0 else 2)
            y2 = conv(xp, doc, d, emb=emb, res=res)
            torch.cuda.synchronize()
            err = relmax(y2, y_ref)
            t2 = timeit(lambda: conv(xp, doc, d, emb=emb, res=res))
            out['%s_v%d_ms' % (prec, variant)] = t2
            out['%s_v%d_err' % (prec, variant)] = err
            out['%s_v%d_TF' % (prec, variant)] = flops / t2 / 1e9
        # fused statistics path against a stand-alone pass over the output
        conv.emit_stats = True
        y3 = conv(xp, doc, d, emb=emb, res=res)
        st = ops.get_stats(y3).view(B, cout, 2).clone()
        conv.emit_stats = False
        bid = doc.batch_id(d)
        s_ref = torch.zeros(B, cout, dtype=torch.float64, device=dev).index_add_(0, bid, y3.double())
        q_ref = torch.zeros(B, cout, dtype=torch.float64, device=dev).index_add_(0, bid, y3.double() ** 2)
        out['%s_stats_err' % prec] = max(float((st[..., 0] - s_ref).abs().max() / s_ref.abs().max()),
                                         float((st[..., 1] - q_ref).abs().max() / q_ref.abs().max()))
        t_split = timeit(lambda: ops.planes_split(x, mode))
        out['%s_split_ms' % prec] = t_split
    ops.set_precision('bf16x3')
    _lib.call('ofx_set_gconv2_tile', 0)
    rows.append(out)
    print(json.dumps(out))
    sys.stdout.flush()

# GroupNorm apply: fp32 output vs planes output (same pass, different store format)
for d, C in [(6, 128), (6, 256)]:
    N = doc.csr(d)[2]
    gn = M.DualOctreeGroupNorm(C).to(dev)
    x = torch.randn(N, C, device=dev)
    y = gn(x, doc, d, act='silu')
    yp = gn(x, doc, d, act='silu', planes=2)
    yh = gn(x, doc, d, act='silu', planes=1)
    t0 = timeit(lambda: gn(x, doc, d, act='silu'))
    t2 = timeit(lambda: gn(x, doc, d, act='silu', planes=2))
    t1 = timeit(lambda: gn(x, doc, d, act='silu', planes=1))
    print(json.dumps(dict(gn_d=d, C=C, fp32_ms=t0, planes2_ms=t2, planes1_ms=t1,
                          planes2_err=relmax(ops.planes_merge(yp, 2), y), planes1_err=relmax(ops.planes_merge(yh, 1), y))))
if args.json:
    json.dump(rows, open(args.json, 'w'), indent=1)
