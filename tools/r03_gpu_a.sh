# round 3, GPU call A: extend the GEMM tile-choice file, new tests, configs[3] bench + kernel stats, shard-of-8 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_a
mkdir -p $O
cd $R
python3 tools/gen_gemm_tune.py > $O/tune.log 2>&1; tail -2 $O/tune.log
cp omchat_amd/gemm_tune_gfx950.txt $O/gemm_tune_gfx950.txt
timeout 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_vit300m.py -x -q > $O/pytest_new.log 2>&1; tail -5 $O/pytest_new.log
python3 bench.py --workload configs3 --steps 3 --warmup 1 > $O/bench_configs3.json 2> $O/bench_configs3.err; head -c 1500 $O/bench_configs3.json; echo
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats3 -o s --output-format csv -- python3 $R/bench.py --workload configs3 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline > $O/stats3.json 2> $O/stats3.err
rocprofv3 --kernel-trace --stats -d $O/shard8_c1 -o s --output-format csv -- python3 $R/bench.py --shard-of 8 --workload configs1 --steps 1 --warmup 1 --gen 32 > $O/shard8_c1.json 2> $O/shard8_c1.err
rocprofv3 --kernel-trace --stats -d $O/shard8_c2 -o s --output-format csv -- python3 $R/bench.py --shard-of 8 --workload configs2 --steps 1 --warmup 1 --gen 16 > $O/shard8_c2.json 2> $O/shard8_c2.err
cd $R
for d in stats3 shard8_c1 shard8_c2; do cp $(find $O/$d -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$d.csv; rm -rf $O/$d; done
python3 bench.py --shard-of 8 --steps 3 --warmup 1 > $O/bench_shard8.json 2> $O/bench_shard8.err; head -c 1200 $O/bench_shard8.json; echo
python3 bench.py --shard-of 4 --steps 2 --warmup 1 > $O/bench_shard4.json 2> $O/bench_shard4.err
python3 bench.py --shard-of 2 --steps 2 --warmup 1 > $O/bench_shard2.json 2> $O/bench_shard2.err
python3 bench.py --steps 3 --warmup 1 > $O/bench_default.json 2> $O/bench_default.err; head -c 600 $O/bench_default.json; echo
tail -3 $O/*.err
