#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_generate.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest a rc=$?"; tail -3 $O/pytest_a.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
bash tools/final_runs.sh
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 timeout 300 python tools/gconv3_timeline.py > gpurun_out/final/gconv3_timeline.txt 2>&1; echo "timeline rc=$?"
timeout 400 python tools/gconv3_ab.py --json gpurun_out/final/gconv3_ab_shell6_b8.json > gpurun_out/final/gconv3_ab.txt 2>&1; echo "ab rc=$?"
