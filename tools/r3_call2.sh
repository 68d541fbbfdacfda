#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so timeout 300 python tools/gconv3_timeline.py > $O/timeline.log 2>&1; echo "timeline rc=$?"
timeout 600 python -m pytest tests/test_gpu_persistent.py -x -q > $O/pytest_persistent.log 2>&1; echo "persistent rc=$?"
tail -3 $O/pytest_persistent.log
cat $O/timeline.log
