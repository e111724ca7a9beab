# round 3, GPU call AG: kernel stats of the e4m3-weight decode block (bench.py fp8_decode) -- never profiled on its own
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_ag
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 64 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1_fp8.csv; rm -rf $O/stats
grep -E "Lb1|true|quant" $O/kernel_stats_configs1_fp8.csv | cut -c1-180
