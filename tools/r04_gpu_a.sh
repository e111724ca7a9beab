# round 4, GPU call A: hand-off microbenchmark (tools/tune_handoff.hip) + a short baseline bench line of the round-3 code
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_a
timeout 600 tools/bin/tune_handoff > gpurun_out/r04_a/handoff.txt 2>&1; echo "handoff rc=$?"; tail -40 gpurun_out/r04_a/handoff.txt
timeout 900 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 > gpurun_out/r04_a/bench_c1.json 2> gpurun_out/r04_a/bench_c1.err; echo "bench rc=$?"
head -c 1500 gpurun_out/r04_a/bench_c1.json
