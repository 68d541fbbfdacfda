#!/bin/bash
# Round-end evidence run on the GPU box: PMC passes over the planes-kernel probe, bench lines of the four BASELINE
# workloads, rocprofv3 kernel stats of the default bench command, training-side probe.  Everything lands under gpurun_out/final/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
# PMC passes first: separate rocprofv3 --pmc runs over the planes-kernel probe (no tracing domains mixed in)
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  n=$(echo $P | cut -d" " -f1)
  timeout -s KILL 90 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc/$n -o p -- python tools/pmc_probe2.py > $OUT/pmc_$n.log 2>&1
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json
# the bench lines below stamp roofline.traffic from this file (it carries the sha of the kernel sources it was measured on)
cp $OUT/pmc_traffic.json profiles/r02/pmc_traffic.json
timeout 300 python bench.py --steps 20 --warmup 5 --layers > $OUT/bench_hr.json 2> $OUT/bench_hr.err
for w in lr hr_cond feature; do
  timeout 300 python bench.py --workload $w --layers > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- \
  python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --eager > $OUT/bench_hr_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/prof -name "*kernel_trace.csv" -delete
timeout 200 python tools/vae_train_probe.py --json $OUT/vae_train_probe.json > $OUT/vae_train_probe.log 2>&1
python - <<'PY'
import json
for w in ('hr', 'lr', 'hr_cond', 'feature'):
    try:
        r = json.load(open('gpurun_out/final/bench_%s.json' % w))
    except Exception as e:
        print(w, 'FAILED', e); continue
    rf = r['roofline']
    print(w, {k: r.get(k) for k in ('value', 'ms_per_step', 'execution', 'eager_ms_per_step', 'fp32_ms_per_step', 'shape_steps_per_s')},
          {k: v['ms_per_step'] for k, v in r.get('side_runs', {}).items()}, 'frac', rf.get('frac'), 'tf', rf.get('algorithmic_TFLOPs'),
          'parity', r.get('parity_spot_check', {}).get('rel_to_max_vs_oracle'), 'cpu', r.get('cpu_baseline', {}).get('value'))
PY
