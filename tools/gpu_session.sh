#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ('value','ttft_ms_p50','vit_ms_p50','prefill_ms_p50','vit_mfma_frac','prefill_mfma_frac','decode_ms_per_token_p50')})
    if 'configs2' in d: print('  configs2', {k:d['configs2'].get(k) for k in ('tokens_per_sec','vit_ms_p50','prefill_ms_p50','decode_ms_per_step_p50','vit_mfma_frac','prefill_mfma_frac','decode_hbm_frac')})
except Exception as e: print('parse fail', sys.argv[1], e)
PY
}
timeout 900 python bench.py --steps 3 --warmup 1 --gen 64 --no-fp8 --no-cpu-baseline > gpurun_out/k_v2.json 2> gpurun_out/k_v2.err; echo "v2 rc=$?"; show gpurun_out/k_v2.json
timeout 900 python bench.py --steps 3 --warmup 1 --gen 64 --no-fp8 --no-cpu-baseline --tuning 8=0 > gpurun_out/k_v1.json 2> gpurun_out/k_v1.err; echo "v1 rc=$?"; show gpurun_out/k_v1.json
timeout 900 python bench.py --steps 3 --warmup 1 --gen 64 --no-fp8 --no-cpu-baseline > gpurun_out/k_v2b.json 2> gpurun_out/k_v2b.err; echo "v2 again rc=$?"; show gpurun_out/k_v2b.json
timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/k_alltests.log 2>&1; echo "all gpu tests rc=$?"
tail -22 gpurun_out/k_alltests.log
