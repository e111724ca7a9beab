#!/bin/bash
# round 6, GPU call I: the N-rank bench path with N rank PROCESSES on one GPU (IPC peer transport, no RCCL; timings meaningless) on the sequence-parallel form:
# TP = N against TP = 1 (tp1_check), counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_i
for n in 2 8; do
  OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 900 python3 bench.py --gpus $n --dtype f16 --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r06_i/bench_os_$n.json 2> gpurun_out/r06_i/bench_os_$n.err
  echo "rc=$?"; python3 -c "
import json; d=json.load(open('gpurun_out/r06_i/bench_os_$n.json')); print($n, {k: d.get(k) for k in ('tokens_match_tp1','tp1_check','transport','comm_stats','peer_timeouts','rccl_nranks')})"
done
