#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ag
for v in nt1 nt0 nt1 nt0; do
  if [ $v = nt0 ]; then export OMCHAT_LIB=$PWD/ab_lib/lib_kvnt0.so; else unset OMCHAT_LIB; fi
  echo "== $v"; python tools/bench_attn_decode.py 1 3640 0 1 2>&1 | grep -v amdgpu.ids | head -1
done
unset OMCHAT_LIB
cd /tmp && export TMPDIR=/tmp
for v in nt1 nt0; do
  if [ $v = nt0 ]; then export OMCHAT_LIB=$GRAFT_REPO_ROOT/ab_lib/lib_kvnt0.so; else unset OMCHAT_LIB; fi
  rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04_ag/p_$v -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_attn_decode.py 1 3640 0 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r04_ag/p_$v -name "*kernel_stats.csv" | head -1); echo "== $v"; head -4 $f | cut -c1-160; rm -rf $GRAFT_REPO_ROOT/gpurun_out/r04_ag/p_$v
done
