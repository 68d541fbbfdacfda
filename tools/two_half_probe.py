"""Does the denoising step gain from running two half-batches on two HIP streams?

The step is a strict chain per shape, but shapes of a batch are independent: while one half-batch sits in an MFMA-bound
GraphConv the other could be in an HBM-bound GroupNorm / gather kernel.  This probe measures the aggregate rate of

  (a) one hipGraph of the whole batch (what bench.py / sampler.sample_loop replay today), against
  (b) two hipGraphs of half the batch each, replayed on two streams at once, the persistent GraphConv planned for
      `--cus` compute units per launch (ofx_set_gconv_cus; 0 = all of them, i.e. both halves ask for the whole chip).

History: the first run of this probe (profiles/r06/two_half_*.txt) showed -12 ... -16 %.  That was a data race on the
then process-wide GroupNorm statistics pool (now per stream): the lanes' activations went non-finite and the launches ran
faster on NaN operands.  The probe now checks that every lane's state is finite; on correct data the gain is 2-5 %
(profiles/r06/lanes_ab.txt, bench.py --lanes 2).

usage: python tools/two_half_probe.py [--workload hr] [--batch 8] [--steps 20] [--cus 0,192,160,128]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

torch.set_grad_enabled(False)


def capture(wl, stream, i0=5):
    """Eager warm-up on `stream` (per-stream scratch must exist before capture), then one captured step."""
    with torch.cuda.stream(stream):
        for i in range(3):
            wl.step(i)
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            wl.step(i0)
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='hr')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--cus', default='0,192,160,128')
    ap.add_argument('--parts', default='2')
    a = ap.parse_args()
    from octfusion_amd import _lib, ops
    dev = torch.device('cuda:0')
    lib = _lib.lib()
    out = {'workload': a.workload, 'batch': a.batch, 'steps': a.steps}

    def time_replays(graphs_streams):
        def go():
            for _ in range(a.steps):
                for g, s in graphs_streams:
                    with torch.cuda.stream(s):
                        g.replay()
            for _, s in graphs_streams:
                s.synchronize()
        go()
        return min(bench.timed(go) for _ in range(3)) / a.steps * 1e3

    s0 = torch.cuda.Stream(dev)
    whole = bench.Workload(a.workload, a.batch, dev, 0, lanes=1)
    g_whole = capture(whole, s0)
    out['whole_ms'] = time_replays([(g_whole, s0)])
    print('whole batch %d: %.3f ms/step' % (a.batch, out['whole_ms']), flush=True)

    out['halves'] = []
    for parts in [int(c) for c in a.parts.split(',')]:
        if a.batch % parts:
            continue
        streams = [torch.cuda.Stream(dev) for _ in range(parts)]
        wls = [bench.Workload(a.workload, a.batch // parts, dev, 0, lanes=1) for _ in range(parts)]
        for w in wls[1:]:
            w.net = wls[0].net
            if wls[0].nested is not None:
                w.nested = wls[0].nested
        for cus in [int(c) for c in a.cus.split(',')]:
            lib.ofx_set_gconv_cus(cus)
            gs = [capture(w, s) for w, s in zip(wls, streams)]
            one = time_replays([(gs[0], streams[0])])
            allp = time_replays(list(zip(gs, streams)))
            rec = {'parts': parts, 'cus': cus, 'part_alone_ms': one, 'all_parts_ms': allp, 'vs_whole': allp / out['whole_ms'],
                   'finite': all(w.finite() for w in wls)}
            out['halves'].append(rec)
            print(json.dumps(rec), flush=True)
            del gs
        del wls
    lib.ofx_set_gconv_cus(0)
    assert not ops.sync_error(dev)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
