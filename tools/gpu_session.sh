#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fp8.py -q -x --durations=5 > gpurun_out/h_tests1.log 2>&1; echo "fp8 tests rc=$?"
tail -25 gpurun_out/h_tests1.log
timeout 900 python tools/gen_gemm_tune.py > gpurun_out/h_tune.log 2>&1; echo "tune rc=$?"; tail -3 gpurun_out/h_tune.log
show() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], {k:d.get(k) for k in ('value','decode_ms_per_token_p50','decode_hbm_frac','ttft_ms_p50','vit_ms_p50','prefill_ms_p50','vit_mfma_frac','prefill_mfma_frac')})
    print(' roofline', d.get('roofline'))
    print(' roofline_prefill', d.get('roofline_prefill'))
except Exception as e: print('parse fail', sys.argv[1], e)
PY
}
timeout 900 python bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > gpurun_out/h_c4.json 2> gpurun_out/h_c4.err; echo "c4 rc=$?"; show gpurun_out/h_c4.json; tail -3 gpurun_out/h_c4.err
bash tools/collect_profiles.sh r02_a > gpurun_out/h_prof.log 2>&1; echo "profiles rc=$?"; tail -16 gpurun_out/h_prof.log
