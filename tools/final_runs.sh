#!/bin/bash
# Round evidence run on the GPU box: PMC passes over the planes-kernel probe, bench lines of the four BASELINE
# workloads, rocprofv3 kernel stats of the default bench command (+ the gather microbenchmark row), the generate probe.
# Everything lands under gpurun_out/final/; copy what is to be judged into profiles/r04/ (tools/collect_profiles.sh).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/final
mkdir -p $OUT
export TMPDIR=/tmp
# PMC passes first: separate rocprofv3 --pmc runs over the planes-kernel probe (no tracing domains mixed in)
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  n=$(echo $P | cut -d" " -f1)
  (cd /tmp && timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OLDPWD/$OUT/pmc/$n -o p -- python $OLDPWD/tools/pmc_probe2.py > $OLDPWD/$OUT/pmc_$n.log 2>&1)
done
python tools/pmc_summary.py $OUT/pmc $OUT/pmc_traffic.json > $OUT/pmc_summary.log 2>&1
# the bench lines below stamp roofline.traffic from this file (it carries the sha of the kernel sources it was measured on)
mkdir -p profiles/r04 && cp $OUT/pmc_traffic.json profiles/r04/pmc_traffic.json
timeout 400 python bench.py --steps 20 --warmup 5 --layers > $OUT/bench_hr.json 2> $OUT/bench_hr.err
for w in lr hr_cond feature; do
  timeout 400 python bench.py --workload $w --layers > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o bench -- \
  python $OLDPWD/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --eager > $OLDPWD/$OUT/bench_hr_under_rocprof.json 2> $OLDPWD/$OUT/rocprof.err)
(cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_gather -o gather -- \
  python $OLDPWD/tools/gather_probe.py > $OLDPWD/$OUT/gather_under_rocprof.json 2> $OLDPWD/$OUT/rocprof_gather.err)
find $OUT/prof $OUT/prof_gather -name "*kernel_trace.csv" -delete
timeout 600 python tools/generate_probe.py --out $OUT/generate_probe.json > $OUT/generate_probe.log 2>&1
timeout 300 python tools/checkpoint_memory_probe.py --out $OUT/checkpoint_memory.json > $OUT/checkpoint_memory.log 2>&1
[ -x tools/probes/mfma_rate ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o tools/probes/mfma_rate tools/probes/mfma_rate.hip
./tools/probes/mfma_rate > $OUT/mfma_rate_probe.txt 2>&1
for w in hr lr; do python tools/step_trace.py --workload $w --out $OUT/step_trace_$w.json > /dev/null 2>&1; done
python tools/step_trace.py --workload hr --batch 1 --out $OUT/step_trace_hr_b1.json > /dev/null 2>&1
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
