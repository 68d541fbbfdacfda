#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persistent.py tests/test_gpu_fullwidth.py -m gpu -x -q -k "persistent or edge or properties or labels" > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
timeout 400 python tools/gconv3_ab.py --json $O/gconv3_ab.json > $O/gconv3_ab.txt 2>&1; echo "ab rc=$?"
tail -1 $O/gconv3_ab.txt
OFX_LIB=$PWD/octfusion_amd/libofx_ablation.so G3_TILES=4 G3_RAW=$O/raw timeout 300 python tools/gconv3_timeline.py > $O/gconv3_timeline.txt 2>&1; echo "timeline rc=$?"
grep -v amdgpu $O/gconv3_timeline.txt
