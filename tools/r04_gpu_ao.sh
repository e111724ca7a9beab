#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_ops.py -q -x -k "bits or gemv" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
python tools/bench_gemv_b1.py 2>&1 | grep -v amdgpu.ids
python tools/bench_gemv_b1.py 2>&1 | grep -v amdgpu.ids
