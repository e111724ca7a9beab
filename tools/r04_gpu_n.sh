#!/bin/bash
# GPU call N: per-XCD share experiment on the batch-1 gate|up GEMV
mkdir -p gpurun_out/r04_n
python tools/bench_gemv_skew.py > gpurun_out/r04_n/skew.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_n/skew.log
