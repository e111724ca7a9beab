#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ac
for L in 8800 16384; do
python tools/bench_attn_decode.py 1 $L 0 0 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r04_ac/a.log
python tools/bench_attn_decode.py 1 $L 0 2 2>&1 | grep -v amdgpu.ids | sed -n 2,4p >> gpurun_out/r04_ac/a.log
done
python tools/bench_attn_decode.py 2 8800 0 0 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r04_ac/a.log
python tools/bench_attn_decode.py 2 8800 0 2 2>&1 | grep -v amdgpu.ids | sed -n 2,4p >> gpurun_out/r04_ac/a.log
cat gpurun_out/r04_ac/a.log
