#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" > gpurun_out/m_tests1.log 2>&1; echo "gemm tests rc=$?"
tail -5 gpurun_out/m_tests1.log
timeout 600 python tools/bench_gemm.py 2,10,8 2>&1 | grep -v amdgpu
