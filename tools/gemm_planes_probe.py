"""The unpool GEMM: register-staged kernel (ofx_gemm_f32_planes) vs the planes data path (ofx_gemm_planes) incl. its
gather + split of the M input rows, per shape of the hr / feature workloads (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import ops, modules as M
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
mode = ops.planes_mode()
for Nd, Mr, C in [(32768, 4976, 512), (67600, 21344, 256), (217008, 71088, 256), (714624, 361968, 128)]:
    up = M.Upsample(C).to(dev)
    x = torch.randn(Nd, C, device=dev)
    a_rows = torch.sort(torch.randperm(Nd, device=dev)[:Mr])[0].to(torch.int32)
    out = torch.empty(Mr, 8 * C, device=dev)
    res = {}
    for on in (False, True):
        ops.GEMM_PLANES = on
        res[on] = timeit(lambda: up(x, a_rows=a_rows, out=out, out_planes=mode))
    fl = 2.0 * Mr * C * 8 * C
    xa = torch.empty(Mr, C, device=dev)
    t_copy = timeit(lambda: ops.rows_copy(x, xa, Mr, smap=a_rows, planes=mode))
    setattr(xa, ops.PLANES_ATTR, mode)
    pgp = up._pgp.get(up.weights.view(C, 8 * C), mode)
    t_pl = timeit(lambda: ops.gemm_planes(xa, pgp, out, mode)) if C >= 256 else float('nan')
    t_f32 = timeit(lambda: ops.gemm_planes(xa, pgp, out, 0)) if C >= 256 else float('nan')
    t_old_f32 = timeit(lambda: ops.gemm(x, up.packed(), out=out, a_rows=a_rows))
    print('unpool M=%d K=%d N=%d: register-staged %.1f us (%.0f TF/s; fp32 out %.1f) | planes path %.1f us (%.0f TF/s) = gather+split %.1f + '
          'GEMM %.1f (fp32 out: %.1f)' % (Mr, C, 8 * C, res[False], fl / res[False] / 1e6, t_old_f32, res[True], fl / res[True] / 1e6,
                                          t_copy, t_pl, t_f32), flush=True)
