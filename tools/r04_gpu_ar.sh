#!/bin/bash
# GPU call AR: re-measure every GEMM tile choice on the current kernels, then TTFT with the committed file against the fresh one
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ar
cp omchat_amd/gemm_tune_gfx950.txt gpurun_out/r04_ar/tune_committed.txt
OMCHAT_GEMM_TUNE_FILE=/nonexistent python tools/gen_gemm_tune.py 2>&1 | grep -v amdgpu.ids | tail -3
cp gpurun_out/gemm_tune_gfx950.txt gpurun_out/r04_ar/tune_fresh.txt
cp gpurun_out/r04_ar/tune_committed.txt omchat_amd/gemm_tune_gfx950.txt
for v in committed fresh committed fresh; do
  OMCHAT_GEMM_TUNE_FILE=$PWD/gpurun_out/r04_ar/tune_$v.txt python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v configs1', 'ttft %.2f vit %.2f prefill %.2f tune_runs %s' % (d['ttft_ms_p50'], d['vit_ms_p50'], d['prefill_ms_p50'], d.get('gemm_tune_measurements')))"
done
for v in committed fresh; do
  OMCHAT_GEMM_TUNE_FILE=$PWD/gpurun_out/r04_ar/tune_$v.txt python3 bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v configs2', 'value %.1f ttft %.1f vit %.1f prefill %.1f' % (d['value'], d['ttft_ms_p50'], d['vit_ms_p50'], d['prefill_ms_p50']))"
done
diff gpurun_out/r04_ar/tune_committed.txt gpurun_out/r04_ar/tune_fresh.txt | grep "^[<>]" | wc -l
