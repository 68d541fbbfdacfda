#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3f; mkdir -p $O
timeout 900 python tools/precision_attribution.py --out $O/precision_attribution.json > $O/precision_attribution.log 2>&1; echo "attribution rc=$?"
timeout 1500 python -m pytest tests/test_gpu_precision.py tests/test_gpu_persistent.py -q > $O/pytest_a.log 2>&1; echo "a rc=$?"
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullwidth.py -q -x > $O/pytest_b.log 2>&1; echo "b rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench hr rc=$?"
timeout 400 python bench.py --workload lr --no-cpu-baseline > $O/bench_lr.json 2> $O/bench_lr.err; echo "bench lr rc=$?"
tail -4 $O/pytest_a.log; tail -4 $O/pytest_b.log; grep -h "setting" $O/precision_attribution.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if 'vs_fp64_oracle' in r:
        print('%-8s %-12s %-44s eager %.2f ms  p99.9 %.2e  rel-to-max %.2e' % (r['step'], r['config'], r['setting'], r['eager_ms'], r['vs_fp64_oracle']['elementwise_p999'], r['vs_fp64_oracle']['rel_to_max']))
    else:
        print('%-8s %-12s %-44s                 p99.9 %.2e  rel-to-max %.2e' % (r['step'], r['config'], r['setting'], r['elementwise_p999'], r['rel_to_max']))
"
