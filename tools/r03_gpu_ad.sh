# round 3, GPU call AD: kernel stats of a warmed configs[4] step (32-frame clip, 33 k context, fp8 modes), never profiled before
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ad
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs4 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs4.csv; rm -rf $O/stats
head -24 $O/kernel_stats_configs4.csv | cut -c1-170
