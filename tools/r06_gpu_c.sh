#!/bin/bash
# round 6, GPU call C: kernel stats of the ViT with the fused layer on / off (tuning key 44), remaining new tests
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_c
timeout 900 python3 -m pytest tests/test_gpu_round6.py -q -x -k "statistics or stats_finish or qk_norm or fused_vit" 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_c; cd $R
for v in 1 0; do
  rm -rf $O/p
  rocprofv3 --kernel-trace --stats -d $O/p -o t --output-format csv -- python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 4 --no-cpu-baseline --no-side --no-fp8 --tuning 44=$v > $O/bench_44_$v.json 2> $O/bench_44_$v.err
  cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats_44_$v.csv
  echo "== key 44 = $v"; head -22 $O/kernel_stats_44_$v.csv | cut -c1-150
  rm -rf $O/p
done
