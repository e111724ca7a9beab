#!/bin/bash
# GPU call Y: prefill attention with the transposed V reads as inline assembly (no compiler vmcnt(0) in the PV phase) against the twin build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_y
python -m pytest tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -x -k "attn_prefill or mha or golden or 25_head or head_dim_64 or 300m" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -4
for v in asm1 asm0; do
  if [ $v = asm0 ]; then export OMCHAT_LIB=$PWD/ab_lib/lib_trasm0.so; else unset OMCHAT_LIB; fi
  echo "== $v" >> gpurun_out/r04_y/attn.log
  python tools/bench_attn_ab.py 27 0 20 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_y/attn.log
done
unset OMCHAT_LIB
cat gpurun_out/r04_y/attn.log
