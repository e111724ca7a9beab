#!/bin/bash
# round 6, GPU call K: kernel stats of the ViT attention with the odd key folded into the initial state (tuning key 46) on / off, same box
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_k; mkdir -p $O; cd $R
for v in 1 0 1 0; do
  rm -rf $O/p
  rocprofv3 --kernel-trace --stats -d $O/p -o t --output-format csv -- python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 4 --no-cpu-baseline --no-side --no-fp8 --tuning 46=$v > /dev/null 2> $O/err_$v.txt
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  echo "key 46 = $v: $(grep 'attn2_kernelIDF16bLi4ELb0ELi128ELi1E' $f | cut -d, -f2-4)"
  rm -rf $O/p
done
