#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$?"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -6 $O/pytest_all.log; tail -5 $O/smoke.log
bash tools/final_runs.sh
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 timeout 300 python tools/gconv3_timeline.py > gpurun_out/final/gconv3_timeline.txt 2>&1; echo "timeline rc=$?"
