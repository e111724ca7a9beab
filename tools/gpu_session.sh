#!/bin/bash
# scratch driver for one gpurun call (rewritten per experiment): the round's closing verification
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -a "passed\|failed" gpurun_out/full_gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1200 bash tools/collect_profiles.sh r02_d > gpurun_out/collect.log 2>&1; echo "collect rc=$?"
head -c 600 gpurun_out/r02_d/bench.json; echo
