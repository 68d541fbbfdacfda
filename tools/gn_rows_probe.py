"""One-launch GroupNorm of the dense grids: per-group blocks (knob 0) vs 16-channel blocks (knob 1), us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from octfusion_amd import _lib, ops

dev = torch.device('cuda:0')
torch.set_grad_enabled(False)
for B, rows, C in [(4, 4096, 64), (4, 512, 128), (4, 64, 256), (8, 4096, 64), (8, 512, 128), (8, 512, 256), (8, 4096, 128)]:
    x = torch.randn(B * rows, C, device=dev)
    w = torch.randn(C, device=dev)
    b = torch.randn(C, device=dev)
    bid = torch.arange(B, device=dev).repeat_interleave(rows).to(torch.int32)
    cnt = torch.full((B,), float(rows), device=dev)
    out = torch.empty_like(x)
    line = 'B=%d rows=%d C=%d:' % (B, rows, C)
    for knob in (0, 1, 2):
        _lib.call('ofx_set_gn_rows16', knob & 1)
        rpb = rows if knob < 2 else None                      # 2: stats + finalize + apply (three launches)
        fn = lambda: ops.group_norm(x, bid, cnt, B, w, b, 32, act='silu', out=out, count_eps=0.0, rows_per_batch=rpb)   # noqa: E731
        for _ in range(5):
            fn()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(50):
                fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        line += '  knob%d %.2f us' % (knob, e0.elapsed_time(e1) * 1e3 / 200)
    print(line)
_lib.call('ofx_set_gn_rows16', 1)
