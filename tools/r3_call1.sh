#!/bin/bash
# round-3 GPU call 1: correctness of the persistent stream-K kernel, then A/B timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py -x -q > $O/pytest_persistent.log 2>&1; echo "persistent rc=$?" | tee -a $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_fullwidth.py -x -q -k "edge_cases" > $O/pytest_edge.log 2>&1; echo "edge rc=$?" | tee -a $O/rc.txt
timeout 400 python tools/gconv3_ab.py --json $O/ab_shell6_b8.json > $O/ab_shell6_b8.log 2>&1; echo "ab rc=$?" | tee -a $O/rc.txt
timeout 400 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench rc=$?" | tee -a $O/rc.txt
tail -5 $O/pytest_persistent.log; tail -5 $O/pytest_edge.log; tail -3 $O/ab_shell6_b8.log
