#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round4.py tests/test_gpu_ops.py tests/test_gpu_round3.py tests/test_gpu_fp8.py -q -x -k "bits or gemv or decode or norm" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
python tools/bench_gemv_b1.py 2>&1 | grep -v amdgpu.ids
R=$GRAFT_REPO_ROOT
for v in r04 r03 r04 r03; do
  if [ $v = r03 ]; then D=$R/ab_lib/r03_tree; else D=$R; fi
  (cd $D && python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'value %.1f decode %.4f ttft %.2f  fp8 decode %s' % (d['value'], d['decode_ms_per_token_p50'], d['ttft_ms_p50'], (d.get('fp8_decode') or {}).get('decode_ms_per_token')))")
done
