#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attn_decode" > gpurun_out/m_tests1.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/m_tests1.log
for t in 1 2 4 0; do
  echo "--- configs2 tpw=$t"
  timeout 900 python bench.py --no-cpu-baseline --workload configs2 --steps 1 --warmup 1 --tuning 10=$t 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in d if 'decode' in k and not isinstance(d[k],dict)}, d['value'])"
done
