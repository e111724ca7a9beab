#!/bin/bash
# GPU call M: does the batched decode attention camp on HBM channels?  cache strides (cap) and rotated tile order
mkdir -p gpurun_out/r04_m
for cap in 3840 4096 3848 3720 3704; do
  python tools/bench_attn_decode.py 32 3700 $cap >> gpurun_out/r04_m/bench_attn.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r04_m/bench_attn.log
