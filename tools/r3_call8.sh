#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py -q -x > $O/pytest_persistent.log 2>&1; echo "persistent rc=$?"
timeout 400 python tools/gconv3_ab.py --json $O/ab_shell6_b8.json > $O/ab_shell6_b8.log 2>&1; echo "ab rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --layers > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench rc=$?"
for P in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OLDPWD/$O/pmc/$P -o p -- python $OLDPWD/tools/pmc_probe2.py > $OLDPWD/$O/pmc_$P.log 2>&1)
done
python - <<'PY'
import csv, glob
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('gpurun_out/r3h/pmc/%s/*counter_collection.csv' % c):
        rows = [r for r in csv.DictReader(open(f)) if 'gconv3' in r['Kernel_Name']]
        vals = [float(r['Counter_Value']) for r in rows]
        print(c, len(vals), [round(sum(vals[i:i + 4]) / 4 / 1024, 1) for i in range(0, len(vals), 4)], 'MB (raw KB/1024) per launch, 4 layers')
PY
tail -3 $O/pytest_persistent.log; tail -1 $O/ab_shell6_b8.log
python -c "
import json; r = json.load(open('$O/bench_hr.json')); print(r['ms_per_step'], r['roofline']['frac'], {k: (round(v['ms_per_step'], 3), round(v['graphconv_mfma_frac'], 3)) for k, v in r['side_runs'].items()})"
