#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ae
for v in new old new old; do
  if [ $v = old ]; then export OMCHAT_LIB=$PWD/ab_lib/lib_gemm_old.so; else unset OMCHAT_LIB; fi
  echo "== $v"; python tools/bench_gemm_ragged.py 2>&1 | grep -v amdgpu.ids | grep "M=3075" | grep -v "tile 2"
done
