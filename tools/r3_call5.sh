#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r3e; mkdir -p $O
timeout 900 python tools/precision_attribution.py --out $O/precision_attribution.json > $O/precision_attribution.log 2>&1; echo "attribution rc=$?"
timeout 1500 python -m pytest tests/test_gpu_precision.py tests/test_gpu_generate.py tests/test_gpu_points.py -q > $O/pytest_new1.log 2>&1; echo "new1 rc=$?"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullwidth.py -q -k "dense_blocks or dense_and_unet or batch8 or obja_hr or lr_step or modules" > $O/pytest_new2.log 2>&1; echo "new2 rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench hr rc=$?"
timeout 400 python bench.py --workload lr --no-cpu-baseline > $O/bench_lr.json 2> $O/bench_lr.err; echo "bench lr rc=$?"
timeout 600 python tools/generate_probe.py --out $O/generate_probe.json > $O/generate_probe.log 2>&1; echo "generate rc=$?"
cd /tmp && timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_hr -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --eager > $GRAFT_REPO_ROOT/$O/bench_hr_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof_hr.err; echo "rocprof hr rc=$?"
cd $GRAFT_REPO_ROOT; find $O/prof_hr -name "*kernel_trace.csv" -delete
tail -4 $O/pytest_new1.log; tail -4 $O/pytest_new2.log; tail -12 $O/precision_attribution.log; tail -3 $O/generate_probe.log
