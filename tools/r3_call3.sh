#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q > $O/pytest_persistent.log 2>&1; echo "persistent rc=$?"
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 timeout 300 python tools/gconv3_timeline.py > $O/timeline.log 2>&1; echo "timeline rc=$?"
timeout 400 python bench.py --steps 20 --warmup 5 --layers --no-cpu-baseline > $O/bench_hr.json 2> $O/bench_hr.err; echo "bench rc=$?"
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all rc=$?"
tail -3 $O/pytest_persistent.log; tail -5 $O/pytest_all.log
grep -v "^   start" $O/timeline.log
