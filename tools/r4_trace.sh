#!/bin/bash
# round-4 early measurement: ordered per-call traces + rocprofv3 kernel trace of the hr step
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/r4a
mkdir -p $OUT
export TMPDIR=/tmp
python tools/step_trace.py --workload hr --out $OUT/step_trace_hr.json > $OUT/st_hr.log 2>&1
python tools/step_trace.py --workload hr --batch 1 --out $OUT/step_trace_hr_b1.json > $OUT/st_hr1.log 2>&1
python tools/step_trace.py --workload lr --out $OUT/step_trace_lr.json > $OUT/st_lr.log 2>&1
(cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o hr -- \
  python $OLDPWD/tools/step_trace.py --workload hr --steps 3 > $OLDPWD/$OUT/prof_hr.log 2>&1)
# keep only the last ~900 rows of the kernel trace (three steady steps)
for f in $(find $OUT/prof -name "*kernel_trace.csv"); do (head -1 $f; tail -n 900 $f) > $OUT/kernel_trace_tail.csv; rm $f; done
tail -3 $OUT/st_hr.log $OUT/st_hr1.log $OUT/st_lr.log
