# round 3, GPU call B: transposed-accumulator GEMM epilogue: full GPU test suite, bench, shard-of-8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_b
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $O/bench_n1.json 2> $O/bench_n1.err; head -c 1500 $O/bench_n1.json; echo
python3 bench.py --shard-of 8 --steps 2 --warmup 1 > $O/bench_shard8.json 2> $O/bench_shard8.err; head -c 1200 $O/bench_shard8.json; echo
