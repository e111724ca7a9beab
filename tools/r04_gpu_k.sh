#!/bin/bash
# GPU call K: the LDS-DMA ring form of the batched decode attention: parity tests, then the kernel-level A/B at b = 32 / 16, then configs2 decode A/B
mkdir -p gpurun_out/r04_k
python -m pytest tests/test_gpu_round4.py -q -x -k "dma_ring" 2>&1 | tail -5 > gpurun_out/r04_k/test.log
python -m pytest tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -x -k "attn_decode or tiles_per_wave or k_through_lds or batched_decode" 2>&1 | tail -5 >> gpurun_out/r04_k/test.log
python tools/bench_attn_decode.py 32 3700 > gpurun_out/r04_k/bench_attn.log 2>&1
python tools/bench_attn_decode.py 16 3700 >> gpurun_out/r04_k/bench_attn.log 2>&1
python tools/bench_attn_decode.py 32 1024 >> gpurun_out/r04_k/bench_attn.log 2>&1
for k in 0 1; do
  python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side --tuning 25=$k > gpurun_out/r04_k/bench_c2_dma$k.json 2> gpurun_out/r04_k/bench_c2_dma$k.err
done
cat gpurun_out/r04_k/test.log gpurun_out/r04_k/bench_attn.log
