#!/bin/bash
# Round evidence run on the GPU box (one box, one call): PMC passes over the planes-kernel probe (hr, hr_cond and feature
# layer sets), bench lines of the four BASELINE workloads, rocprofv3 kernel stats of ALL FOUR bench commands (+ the gather
# microbenchmark row), step traces, native-node listing, the generate probe, and the round's A/B probes (GroupNorm octet
# launch, XCD tile order, dense GEMM on the planes path, input convolution, 1x1 skip tile width).
# Everything lands under gpurun_out/final/; tools/collect_profiles.sh copies what is to be judged into profiles/r06/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
R=r06
OUT=gpurun_out/final
mkdir -p $OUT profiles/$R
export TMPDIR=/tmp
PASSES=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM")
for W in hr hr_cond feature; do
  for P in "${PASSES[@]}"; do
    n=$(echo $P | cut -d" " -f1)
    (cd /tmp && timeout -s KILL 180 rocprofv3 --pmc $P --output-format csv -d $OLDPWD/$OUT/pmc_$W/$n -o p -- python $OLDPWD/tools/pmc_probe2.py $W > $OLDPWD/$OUT/pmc_${W}_$n.log 2>&1)
  done
done
python tools/pmc_summary.py $OUT/pmc_hr $OUT/pmc_traffic.json hr > $OUT/pmc_summary.log 2>&1
python tools/pmc_summary.py $OUT/pmc_hr_cond $OUT/pmc_traffic_hr_cond.json hr_cond >> $OUT/pmc_summary.log 2>&1
python tools/pmc_summary.py $OUT/pmc_feature $OUT/pmc_traffic_feature.json feature >> $OUT/pmc_summary.log 2>&1
# the bench lines below stamp roofline.traffic from these files (they carry the sha of the kernel sources they were measured on)
for f in pmc_traffic.json pmc_traffic_hr_cond.json pmc_traffic_feature.json; do cp $OUT/$f profiles/$R/$f; done
timeout 500 python bench.py --steps 20 --warmup 5 --layers > $OUT/bench_hr.json 2> $OUT/bench_hr.err
for w in lr hr_cond feature; do
  timeout 500 python bench.py --workload $w --layers > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
for w in hr lr hr_cond feature; do
  (cd /tmp && timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$w -o bench -- \
    python $OLDPWD/bench.py --workload $w --steps 10 --warmup 3 --no-extras --no-cpu-baseline --eager > $OLDPWD/$OUT/bench_${w}_under_rocprof.json 2> $OLDPWD/$OUT/rocprof_$w.err)
done
(cd /tmp && timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_gather -o gather -- \
  python $OLDPWD/tools/gather_probe.py > $OLDPWD/$OUT/gather_under_rocprof.json 2> $OLDPWD/$OUT/rocprof_gather.err)
find $OUT -name "*kernel_trace.csv" -delete
for w in hr lr hr_cond feature; do
  python tools/step_trace.py --workload $w --out $OUT/step_trace_$w.json > /dev/null 2>&1
  python tools/native_nodes.py --workload $w --out $OUT/native_nodes_$w.json > $OUT/native_nodes_$w.log 2>&1
done
python tools/step_trace.py --workload hr --batch 1 --out $OUT/step_trace_hr_b1.json > /dev/null 2>&1
timeout 600 python tools/generate_probe.py --out $OUT/generate_probe.json > $OUT/generate_probe.log 2>&1
# ---- the round's A/B probes
timeout 300 python tools/gn_probe_oct.py both --out $OUT/gn_probe_oct.json > $OUT/gn_probe_oct.txt 2>&1
timeout 300 python tools/gn_probe_oct_parts.py both > $OUT/gn_probe_oct_parts.txt 2>&1
timeout 300 python tools/gconv3_xcd_probe.py shell6 --json $OUT/xcd_order_shell6.json > $OUT/xcd_order_shell6.txt 2>&1
timeout 300 python tools/gconv3_xcd_probe.py shell8 --json $OUT/xcd_order_shell8.json > $OUT/xcd_order_shell8.txt 2>&1
timeout 300 python tools/gemm_planes_probe.py > $OUT/gemm_planes_probe.txt 2>&1
timeout 300 python tools/narrow_in_probe.py > $OUT/narrow_in_probe.txt 2>&1
timeout 300 python tools/skip_gemm_probe.py > $OUT/skip_gemm_probe.txt 2>&1
python - <<'PY'
import json
for w in ('hr', 'lr', 'hr_cond', 'feature'):
    try:
        r = json.load(open('gpurun_out/final/bench_%s.json' % w))
    except Exception as e:
        print(w, 'FAILED', e); continue
    rf = r['roofline']
    print(w, {k: r.get(k) for k in ('value', 'ms_per_step', 'execution', 'eager_ms_per_step', 'fp32_ms_per_step', 'shape_steps_per_s')},
          {k: v['ms_per_step'] for k, v in r.get('side_runs', {}).items()}, 'bound', rf.get('bound'), 'frac', rf.get('frac'), 'tf', rf.get('algorithmic_TFLOPs'),
          'tail', r.get('roofline_tail', {}).get('ms_per_step'),
          'parity', r.get('parity_spot_check', {}).get('rel_to_max_vs_oracle'), 'cpu', r.get('cpu_baseline', {}).get('value'))
PY
