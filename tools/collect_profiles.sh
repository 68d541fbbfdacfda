#!/bin/bash
# Copy the evidence of tools/final_runs.sh (merged back under gpurun_out/final/) into the tracked profiles/r04/.
set -u
cd "$(dirname "$0")/.."
S=gpurun_out/final; D=profiles/r04
mkdir -p $D
for w in hr lr hr_cond feature; do [ -s $S/bench_$w.json ] && cp $S/bench_$w.json $D/bench_r04_$w.json; done
[ -s $S/bench_hr_under_rocprof.json ] && cp $S/bench_hr_under_rocprof.json $D/bench_r04_hr_under_rocprof.json
[ -s $S/prof/bench_kernel_stats.csv ] && cp $S/prof/bench_kernel_stats.csv $D/bench_r04_hr_kernel_stats.csv
[ -s $S/prof_gather/gather_kernel_stats.csv ] && cp $S/prof_gather/gather_kernel_stats.csv $D/gather_r04_kernel_stats.csv
[ -s $S/gather_under_rocprof.json ] && cp $S/gather_under_rocprof.json $D/gather_r04_under_rocprof.json
for n in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT; do
  f=$(ls $S/pmc/$n/*counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $D/pmc_${n}_probe2.csv
done
for f in pmc_traffic.json generate_probe.json checkpoint_memory.json mfma_rate_probe.txt step_trace_hr.json step_trace_lr.json step_trace_hr_b1.json; do
  [ -s $S/$f ] && cp $S/$f $D/$f
done
[ -s gpurun_out/fullwidth_parity.jsonl ] && cp gpurun_out/fullwidth_parity.jsonl $D/fullwidth_parity.jsonl
ls -la $D
