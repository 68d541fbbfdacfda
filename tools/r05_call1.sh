#!/bin/bash
# round-5 GPU call 1: full GPU suite with durations, native-node trace, hr bench + A/B of the new switches
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05a
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q --durations=30 > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -45 $OUT/pytest.log
timeout 200 python tools/native_nodes.py --workload hr --out $OUT/native_nodes_hr.json > $OUT/native_nodes_hr.log 2>&1
tail -60 $OUT/native_nodes_hr.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_hr.json 2> $OUT/bench_hr.err
python - <<'PY'
import json
r = json.load(open('gpurun_out/r05a/bench_hr.json'))
print('hr', r['ms_per_step'], 'eager', r['eager_ms_per_step'], 'frac', r['roofline'].get('frac'), 'tail', r.get('roofline_tail', {}).get('ms_per_step'),
      {k: v['ms_per_step'] for k, v in r.get('side_runs', {}).items()})
PY
for v in "OFX_FORK=0" "OFX_NARROW_IN=0 OFX_NARROW_OUT=0" "OFX_FORK=0 OFX_NARROW_IN=0 OFX_NARROW_OUT=0"; do
  env $v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['eager_ms_per_step'])"
done
timeout 100 python tools/step_trace.py --workload hr --out $OUT/step_trace_hr.json > /dev/null 2>&1
