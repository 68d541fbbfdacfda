#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_fullwidth.py -x -q -k "persistent or edge_cases" > $O/pytest_persistent.log 2>&1; echo "persistent rc=$?"
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 timeout 300 python tools/gconv3_timeline.py > $O/timeline.log 2>&1; echo "timeline rc=$?"
timeout 400 python tools/gconv3_ab.py --json $O/ab_shell6_b8.json > $O/ab_shell6_b8.log 2>&1; echo "ab rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench rc=$?"
tail -3 $O/pytest_persistent.log
grep -v "^   start" $O/timeline.log
tail -1 $O/ab_shell6_b8.log
