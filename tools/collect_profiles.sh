#!/bin/bash
# Copy the evidence of tools/final_runs.sh (merged back under gpurun_out/final/) into the tracked profiles/r06/.
set -u
cd "$(dirname "$0")/.."
R=r06
S=gpurun_out/final; D=profiles/$R
mkdir -p $D
for w in hr lr hr_cond feature; do
  [ -s $S/bench_$w.json ] && cp $S/bench_$w.json $D/bench_${R}_$w.json
  [ -s $S/bench_${w}_under_rocprof.json ] && cp $S/bench_${w}_under_rocprof.json $D/bench_${R}_${w}_under_rocprof.json
  f=$(ls $S/prof_$w/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/bench_${R}_${w}_kernel_stats.csv
  [ -s $S/step_trace_$w.json ] && cp $S/step_trace_$w.json $D/step_trace_$w.json
  [ -s $S/native_nodes_$w.json ] && cp $S/native_nodes_$w.json $D/native_nodes_$w.json
done
f=$(ls $S/prof_gather/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/gather_${R}_kernel_stats.csv
[ -s $S/gather_under_rocprof.json ] && cp $S/gather_under_rocprof.json $D/gather_${R}_under_rocprof.json
for W in hr hr_cond feature; do
  sfx=""; [ $W != hr ] && sfx="_$W"
  for n in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT; do
    f=$(ls $S/pmc_$W/$n/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/pmc_${n}_probe2$sfx.csv
  done
done
for f in pmc_traffic.json pmc_traffic_hr_cond.json pmc_traffic_feature.json generate_probe.json step_trace_hr_b1.json \
         gn_probe_oct.json gn_probe_oct.txt gn_probe_oct_parts.txt xcd_order_shell6.json xcd_order_shell8.json \
         gemm_planes_probe.txt narrow_in_probe.txt skip_gemm_probe.txt; do
  [ -s $S/$f ] && cp $S/$f $D/$f
done
[ -s gpurun_out/fullwidth_parity.jsonl ] && cp gpurun_out/fullwidth_parity.jsonl $D/fullwidth_parity.jsonl
ls -la $D
