#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r05d
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_narrow.py tests/test_gpu_fullwidth.py tests/test_gpu_precision.py -q -x 2>&1 | tail -8
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('hr', r['ms_per_step'], r['eager_ms_per_step'])"
timeout 200 python tools/native_nodes.py --workload hr --out $OUT/native_nodes_hr.json 2>&1 | grep -E "other|src|nodes" | head -20
timeout 200 python bench.py --workload lr --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('lr', r['ms_per_step'], r['eager_ms_per_step'])"
timeout 200 python bench.py --workload hr_cond --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('hr_cond', r['ms_per_step'], r['eager_ms_per_step'])"
timeout 300 python bench.py --workload feature --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('feature', r['ms_per_step'], r['eager_ms_per_step'])"
