#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in new old new old; do
  if [ $v = old ]; then export OMCHAT_LIB=$PWD/ab_lib/lib_gemv_old.so; else unset OMCHAT_LIB; fi
  echo "== $v"; python tools/bench_gemv_b1.py 2>&1 | grep -v amdgpu.ids
done
