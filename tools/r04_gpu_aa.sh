#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_aa
python tools/bench_gemm_sk.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_aa/gemm_sk.log; cat gpurun_out/r04_aa/gemm_sk.log
