#!/bin/bash
# round-2 GPU session B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_peer.py tests/test_gpu_round2.py tests/test_gpu_api.py -x -q --durations=8 > gpurun_out/b_tests1.log 2>&1; echo "tests1 rc=$?"
tail -25 gpurun_out/b_tests1.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q --durations=8 -k "configs1 or batch32" -s > gpurun_out/b_tests2.log 2>&1; echo "tests2 rc=$?"
tail -15 gpurun_out/b_tests2.log
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 300 python bench.py --gpus 2 --tiny --steps 1 --warmup 0 --gen 8 --text-tokens 8 --workload configs1 > gpurun_out/b_tiny2.json 2> gpurun_out/b_tiny2.err; echo "tiny2 rc=$?"
tail -c 1200 gpurun_out/b_tiny2.json; tail -3 gpurun_out/b_tiny2.err
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 1 --gen 40 --workload configs1 > gpurun_out/b_full2.json 2> gpurun_out/b_full2.err; echo "full2 rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/b_full2.json').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ('n_gpus','value','tokens_match_tp1','tp1_check','transport','comm_stats')})
except Exception as e: print('parse fail', e)
PY
tail -3 gpurun_out/b_full2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/b_prof_c2 -- python3 $OLDPWD/bench.py --workload configs2 --steps 1 --warmup 1 --gen 24 --no-fp8 --no-cpu-baseline > $OLDPWD/gpurun_out/b_prof_c2.json 2> $OLDPWD/gpurun_out/b_prof_c2.err; echo "prof rc=$?"
cd $OLDPWD
find gpurun_out/b_prof_c2 -name "*kernel_stats.csv" | head -1 | xargs -I{} head -40 {}
