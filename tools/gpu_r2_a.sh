#!/bin/bash
# round-2 GPU session A: peer all-reduce tests, oversubscribed 2-rank bench (one GPU), N=1 bench with both workloads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_peer.py tests/test_gpu_tp_single.py -x -q > gpurun_out/a_peer_tests.log 2>&1; echo "peer tests rc=$?"
tail -5 gpurun_out/a_peer_tests.log
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 300 python bench.py --gpus 2 --tiny --steps 1 --warmup 0 --gen 8 --text-tokens 8 --workload configs1 > gpurun_out/a_tiny2.json 2> gpurun_out/a_tiny2.err; echo "tiny2 rc=$?"
tail -c 1500 gpurun_out/a_tiny2.json; tail -5 gpurun_out/a_tiny2.err
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --gen 40 --workload configs1 > gpurun_out/a_full2.json 2> gpurun_out/a_full2.err; echo "full2 rc=$?"
tail -c 2500 gpurun_out/a_full2.json; tail -5 gpurun_out/a_full2.err
timeout 900 python bench.py --steps 2 --warmup 1 > gpurun_out/a_bench1.json 2> gpurun_out/a_bench1.err; echo "bench1 rc=$?"
tail -c 6000 gpurun_out/a_bench1.json; tail -5 gpurun_out/a_bench1.err
