#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3j; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
tail -5 $O/smoke.log
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 G3_RAW=$O/raw timeout 300 python tools/gconv3_timeline.py > $O/gconv3_timeline.txt 2>&1; echo "timeline rc=$?"
tail -4 $O/gconv3_timeline.txt
