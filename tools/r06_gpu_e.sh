#!/bin/bash
# round 6, GPU call E: sequence-parallel norms -- TP tests (hook contexts, IPC peer processes), shard lines with the form on / off
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_e
timeout 1200 python3 -m pytest tests/test_gpu_tp_single.py tests/test_gpu_peer.py tests/test_gpu_round3.py -q -x -k "tp or peer" --durations=5 2>&1 | tail -15
for n in 8 4 2; do
  for v in 1 0; do
    python3 bench.py --shard-of $n --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-fp8 --tuning 45=$v > gpurun_out/r06_e/shard${n}_sp$v.json 2> gpurun_out/r06_e/shard${n}_sp$v.err
    python3 -c "
import json; d=json.load(open('gpurun_out/r06_e/shard${n}_sp$v.json')); c=d.get('configs2') or {}
print('shard-of $n sp=$v:', {k: round(d[k],3) for k in ('vit_ms_p50','prefill_ms_p50','decode_ms_per_token_p50') if k in d}, {k: round(c[k],3) for k in ('vit_ms_p50','prefill_ms_p50','decode_ms_per_step_p50') if k in c}, d.get('comm_stats'))"
  done
done
