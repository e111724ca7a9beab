#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_peer.py tests/test_gpu_round2.py tests/test_gpu_api.py -q --durations=5 > gpurun_out/d_tests1.log 2>&1; echo "tests1 rc=$?"
tail -25 gpurun_out/d_tests1.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_ops.py -q -x > gpurun_out/d_tests2.log 2>&1; echo "tests2 rc=$?"
tail -8 gpurun_out/d_tests2.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ('n_gpus','dtype','value','tokens_match_tp1','tp1_check','comm_stats')})
    if 'configs2' in d: print('configs2', {k:d['configs2'].get(k) for k in ('tokens_per_sec','samples_per_sec','decode_ms_per_step_p50','decode_hbm_frac','vit_mfma_frac','prefill_mfma_frac')}, d['configs2']['roofline'])
except Exception as e: print('parse fail', sys.argv[1], e)
PY
}
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --steps 1 --warmup 0 --gen 34 --workload configs1 > gpurun_out/d_full2.json 2> gpurun_out/d_full2.err; echo "full2 rc=$?"; show gpurun_out/d_full2.json
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 4 --steps 1 --warmup 0 --gen 34 --workload configs1 > gpurun_out/d_full4.json 2> gpurun_out/d_full4.err; echo "full4 rc=$?"; show gpurun_out/d_full4.json
timeout 900 python bench.py --steps 2 --warmup 1 --no-fp8 --no-cpu-baseline > gpurun_out/d_bench1.json 2> gpurun_out/d_bench1.err; echo "bench1 rc=$?"; show gpurun_out/d_bench1.json
