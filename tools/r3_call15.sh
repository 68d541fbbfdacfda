#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3o; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1; echo "all rc=$?"
tail -6 $O/pytest_all.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
