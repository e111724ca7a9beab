#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_au
python -m pytest tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_model.py -q -x -k "attn_prefill or mha or golden or vit or tower or 300m or head_dim" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -3
python tools/bench_attn_ab.py 33 0,1 30 vit,vit24 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_au/attn.log
for k in 1 0 1 0; do python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side --tuning 33=$k 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('key33=$k', 'ttft %.2f vit %.2f prefill %.2f' % (d['ttft_ms_p50'], d['vit_ms_p50'], d['prefill_ms_p50']))"; done
