#!/bin/bash
# in-run A/B of an experiment library against the product one: tools/r05_ab.sh <variant name> [workloads...]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
name=$1; shift
for w in "${@:-hr}"; do
  for rep in 1 2; do
    timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$w base   ', round(r['ms_per_step'],4), round(r['eager_ms_per_step'],4), round(r['roofline'].get('frac') or 0,4))"
    OFX_LIB=$PWD/octfusion_amd/libofx_$name.so timeout 300 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$w $name', round(r['ms_per_step'],4), round(r['eager_ms_per_step'],4), round(r['roofline'].get('frac') or 0,4))"
  done
done
