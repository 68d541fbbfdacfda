"""1x1 skip convolutions (Conv1x1 of GraphResBlockEmbed, reference modules.py:332-339, 721-728): dense GEMM with 128- vs
64-column tiles (ofx_set_gemm_bn64) on the shapes of the hr / feature steps; GB/s on the operator's bytes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import _lib, ops
dev = torch.device('cuda:0'); torch.set_grad_enabled(False)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
for M, K, N in [(217008, 384, 128), (217008, 256, 128), (67600, 768, 256), (67600, 384, 256), (67600, 128, 256), (714624, 384, 128),
                (714624, 192, 128), (217008, 512, 256), (217008, 384, 256), (3248400, 128, 64)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    pw = ops.PackedWeight().get(w, 'nk')
    out = torch.empty(M, N, device=dev)
    r = {}
    for bn in (0, 128, 256):
        _lib.call('ofx_set_gemm_bn64', bn)
        r[bn] = timeit(lambda: ops.gemm(x, pw, b, out=out))
    _lib.call('ofx_set_gemm_bn64', 0)
    nb = 4.0 * (M * K + M * N)
    print('skip [%d, %d] -> %d: 128-column tiles %.1f us (%.2f TB/s) | 64-column tiles for N <= 128: %.1f us | for N <= 256: %.1f us (%.2f TB/s)' % (
        M, K, N, r[0], nb / r[0] / 1e6, r[128], r[256], nb / r[256] / 1e6), flush=True)
