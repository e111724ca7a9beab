#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -x -k "packed or gemv or batch" > gpurun_out/m_tests1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/m_tests1.log
for t in 1 0 1 0; do
  echo "--- configs2 no_xs=$t"
  timeout 900 python bench.py --no-cpu-baseline --workload configs2 --steps 1 --warmup 1 --tuning 11=$t 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in d if 'decode' in k and not isinstance(d[k],dict)}, d['value'])"
done
