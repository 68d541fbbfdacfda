# Re-take the hr / lr bench lines and the rocprofv3 kernel stats of the default bench command (a subset of final_runs.sh).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/final2
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python bench.py --steps 20 --warmup 5 --layers > $OUT/bench_hr.json 2> $OUT/bench_hr.err
timeout 400 python bench.py --workload lr --layers > $OUT/bench_lr.json 2> $OUT/bench_lr.err
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- \
  python $OLDPWD/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --eager > $OLDPWD/$OUT/bench_hr_under_rocprof.json 2> $OLDPWD/$OUT/rocprof.err)
find $OUT/prof -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
for w in ('hr','lr'):
    r=json.load(open('gpurun_out/final2/bench_%s.json'%w)); print(w, r['ms_per_step'], r['eager_ms_per_step'], r['value'], r['roofline'].get('frac'))
PY
