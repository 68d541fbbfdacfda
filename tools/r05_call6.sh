#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_attention.py -q -x 2>&1 | tail -8
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('hr', r['ms_per_step'], r['eager_ms_per_step'])"
timeout 200 python bench.py --workload lr --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lr', r['ms_per_step'], r['eager_ms_per_step'])"
timeout 100 python tools/step_trace.py --workload hr --out gpurun_out/r05e/step_trace_hr.json > /dev/null 2>&1
python - <<'PY'
import json
t=json.load(open('gpurun_out/r05e/step_trace_hr.json'))
for r in t['last_step']:
    if 'attention' in r['call']: print(r['call'], round(r['ms']*1e3,1), r['meta'][3])
PY
