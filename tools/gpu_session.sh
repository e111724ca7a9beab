#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_fullsize.py -q -x -k "packed or gemv or batch" 2>&1 | tail -2
for t in 1 0 1 0; do
  echo "--- configs2 no_xs=$t"
  timeout 900 python bench.py --no-cpu-baseline --workload configs2 --steps 1 --warmup 1 --tuning 11=$t 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in d if 'decode_ms' in k}, d['value'])"
done
