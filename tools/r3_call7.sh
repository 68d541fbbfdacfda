#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3g; mkdir -p $O
timeout 300 python tools/gn_probe.py > $O/gn_probe.log 2>&1; echo "gn rc=$?"
for t in 128 32 8; do
  OFX_PLANES_MIN_TILES=$t timeout 300 python tools/generate_probe.py --batches 1,2 --out $O/generate_b1_min$t.json > $O/generate_b1_min$t.log 2>&1; echo "gen $t rc=$?"
done
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench rc=$?"
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "modules or norm or gn" > $O/pytest_gn.log 2>&1; echo "gn tests rc=$?"
cat $O/gn_probe.log | grep -v amdgpu; tail -2 $O/pytest_gn.log
for t in 128 32 8; do echo "== min tiles $t"; grep shapes_per_call $O/generate_b1_min$t.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['shapes_per_call'], 's/shape %.3f' % r['seconds_per_shape'], r['ms_per_step'])
"; done
python -c "
import json; r = json.load(open('$O/bench_hr.json')); print(r['ms_per_step'], r['roofline']['frac'], {k: round(v['ms_per_step'], 3) for k, v in r['side_runs'].items()}, r['per_shape_setup'])"
