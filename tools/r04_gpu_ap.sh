#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py tests/test_gpu_round3.py -q -x -k "gemv or batched or packed" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
python tools/bench_decode_batch.py 2>&1 | grep -v amdgpu.ids | tail -8
python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('configs2 value %.1f decode %.4f' % (d['value'], d['decode_ms_per_token_p50']))"
