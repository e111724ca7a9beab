#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ad
python -m pytest tests/test_gpu_round4.py -q -x -k "head_split" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
python tools/bench_attn_ab.py 31 0,1 20 dec,dec2048,dec5000,dec_b4,c3_8704,long33k 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_ad/attn.log
cat gpurun_out/r04_ad/attn.log
