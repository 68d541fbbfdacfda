#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/final_runs.sh
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 timeout 300 python tools/gconv3_timeline.py > gpurun_out/final/gconv3_timeline.txt 2>&1; echo "timeline rc=$?"
timeout 400 python tools/gconv3_ab.py --json gpurun_out/final/gconv3_ab_shell6_b8.json > gpurun_out/final/gconv3_ab.txt 2>&1; echo "ab rc=$?"
