#!/bin/bash
# GPU call O: ring depth x resident waves of the LDS-DMA batched decode attention
mkdir -p gpurun_out/r04_o
python tools/bench_attn_decode.py 32 3700 > gpurun_out/r04_o/bench_attn.log 2>&1
python tools/bench_attn_decode.py 32 3700 >> gpurun_out/r04_o/bench_attn.log 2>&1
python tools/bench_attn_decode.py 16 3700 >> gpurun_out/r04_o/bench_attn.log 2>&1
grep -v amdgpu.ids gpurun_out/r04_o/bench_attn.log
