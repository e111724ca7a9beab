#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -3 gpurun_out/full_gpu_tests.log
timeout 1200 bash tools/collect_profiles.sh r02_b > gpurun_out/collect.log 2>&1; echo "collect rc=$?"
tail -12 gpurun_out/collect.log
