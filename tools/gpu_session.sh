#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -a "passed\|failed" gpurun_out/full_gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_final.json').read().strip().splitlines()[-1])
c=d['configs2']
print(d['value'], d['decode_ms_per_token_p50'], d['vit_ms_p50'], d['prefill_ms_p50'], d['ttft_ms_p50'], c['tokens_per_sec'], c['decode_ms_per_step_p50'], d['cpu_baseline']['value'])
PY
