#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05c
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_narrow.py -q -x 2>&1 | tail -5
for v in "OFX_X=0" "OFX_NARROW_IN=0"; do
  env $v timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$v', r['ms_per_step'], r['eager_ms_per_step'])"
done
timeout 100 python tools/step_trace.py --workload hr --out $OUT/step_trace_hr.json > /dev/null 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/r05c/step_trace_hr.json'))
for r in t['last_step']:
    if 'narrow' in r['call']: print(r['call'], round(r['ms']*1e3,1), r['meta'][3])
PY
