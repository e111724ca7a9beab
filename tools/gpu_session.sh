#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export OMCHAT_BENCH_OVERSUBSCRIBE=1
for n in 2 4; do
timeout 900 python bench.py --gpus $n --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --workload configs1 --transport peer > gpurun_out/bench_os_$n.json 2> gpurun_out/bench_os_$n.err; echo "N=$n rc=$?"
tail -1 gpurun_out/bench_os_$n.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d.get(k) for k in ('n_gpus','value','rccl_nranks','tokens_match_tp1','tp1_logits_rel_err','decode_ms_per_token_p50','transport','peer_allreduces','rccl_allreduces')})"
tail -3 gpurun_out/bench_os_$n.err
done
