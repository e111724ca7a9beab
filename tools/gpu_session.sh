#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_peer.py -q -x > gpurun_out/l_tests1.log 2>&1; echo "peer tests rc=$?"; tail -5 gpurun_out/l_tests1.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ('n_gpus','value','tokens_match_tp1','vit_data_parallel','transport','comm_stats')}, d['config']['parallelism'])
    print('   tp1', d.get('tp1_check'))
except Exception as e: print('parse fail', sys.argv[1], e)
PY
}
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --gen 34 --workload configs1 --vit both > gpurun_out/l_both.json 2> gpurun_out/l_both.err; echo "both rc=$?"; show gpurun_out/l_both.json; tail -3 gpurun_out/l_both.err
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --gen 34 --workload configs1 --vit dp > gpurun_out/l_dp.json 2> gpurun_out/l_dp.err; echo "dp rc=$?"; show gpurun_out/l_dp.json; tail -3 gpurun_out/l_dp.err
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --steps 1 --warmup 1 --gen 8 --batch2 8 --steps2 1 > gpurun_out/l_both2.json 2> gpurun_out/l_both2.err; echo "both workloads rc=$?"; show gpurun_out/l_both2.json; tail -3 gpurun_out/l_both2.err
