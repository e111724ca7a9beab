#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python bench.py --no-cpu-baseline --steps 1 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['gemm_tune_measurements'], d['configs2']['tokens_per_sec'], d['configs2']['decode_ms_per_step_p50'])"
