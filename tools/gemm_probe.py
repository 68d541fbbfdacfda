"""Dense-GEMM timings of the pool / unpool / 1x1 shapes of the hr and feature steps (GPU box): fp32 vs pair-planes output,
with / without row maps.  us per launch (HIP events), algorithmic TFLOP/s, output GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octfusion_amd import ops
dev = torch.device('cuda:0')
torch.set_grad_enabled(False)


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


shapes = [('unpool d5->d6', 21344, 256, 2048), ('pool d6->d5', 21344, 1024, 128), ('unpool d4->d5', 4976, 512, 4096),
          ('pool d5->d4', 4976, 2048, 256), ('skip d6', 217008, 384, 128), ('skip d5', 67600, 768, 256),
          ('feature unpool d7->d8', 361968, 128, 1024), ('feature unpool d6->d7', 71088, 256, 2048)]
for name, M, K, N in shapes:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(K, N, device=dev) * 0.05
    pw = ops.PackedWeight().get(w, 'kn')
    out = torch.empty(M, N, device=dev)
    rows = torch.randperm(M, device=dev).to(torch.int32)
    res = {}
    for label, kw in (('fp32', {}), ('planes', dict(out_planes=3)), ('planes+a_rows', dict(out_planes=3, a_rows=rows)),
                      ('fp32+out_rows', dict(out_rows=rows))):
        res[label] = timeit(lambda: ops.gemm(a, pw, out=out, **kw))
    fl = 2.0 * M * K * N
    print('%-24s M %7d K %5d N %5d: ' % (name, M, K, N) + '  '.join('%s %.1f us (%.0f TF/s, out %.2f TB/s)' % (k, v, fl / v / 1e6, 4e-6 * M * N / v) for k, v in res.items()))
