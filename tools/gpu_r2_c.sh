#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 tools/bin/tune_gemv32 > gpurun_out/c_tune_gemv32.log 2>&1; echo "tune rc=$?"
cat gpurun_out/c_tune_gemv32.log
timeout 600 python -m pytest tests/test_gpu_peer.py -x -q --durations=5 > gpurun_out/c_tests1.log 2>&1; echo "peer tests rc=$?"
tail -12 gpurun_out/c_tests1.log
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_api.py -q --durations=5 > gpurun_out/c_tests2.log 2>&1; echo "tests2 rc=$?"
tail -30 gpurun_out/c_tests2.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "batch32" > gpurun_out/c_tests3.log 2>&1; echo "tests3 rc=$?"
tail -5 gpurun_out/c_tests3.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ('n_gpus','dtype','tokens_match_tp1','tp1_check','comm_stats')})
except Exception as e: print('parse fail', sys.argv[1], e)
PY
}
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --gen 34 --workload configs1 --dtype f16 > gpurun_out/c_full2_f16.json 2> gpurun_out/c_full2_f16.err; echo "f16 rc=$?"; show gpurun_out/c_full2_f16.json
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --gen 34 --workload configs1 --tuning 4=1000000000 > gpurun_out/c_full2_nochunk.json 2> gpurun_out/c_full2_nochunk.err; echo "nochunk rc=$?"; show gpurun_out/c_full2_nochunk.json
