#!/bin/bash
# GPU call S: automatic choice of the decode attention form: attention tests, decode step by batch size, configs2 / configs3 / configs1 lines
mkdir -p gpurun_out/r04_s
python -m pytest tests/test_gpu_round4.py tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_fullsize.py -q -x -k "dma_ring or attn_decode or tiles_per_wave or k_through_lds or batched_decode or batch_32 or batch32 or configs3" 2>&1 | tail -5 > gpurun_out/r04_s/test.log
python tools/bench_decode_batch.py > gpurun_out/r04_s/decode_batch.log 2>&1
python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_s/c2.json 2> gpurun_out/r04_s/c2.err
python bench.py --workload configs3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_s/c3.json 2> gpurun_out/r04_s/c3.err
python bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_s/c1.json 2> gpurun_out/r04_s/c1.err
cat gpurun_out/r04_s/test.log; grep -v amdgpu.ids gpurun_out/r04_s/decode_batch.log | tail -20
python - <<'PY'
import json
for c in ("c1", "c2", "c3"):
    try:
        d = json.load(open(f"gpurun_out/r04_s/{c}.json")); print(c, "value %.1f" % d["value"], "decode ms %.4f" % d.get("decode_ms_per_token_p50", -1))
    except Exception as e: print(c, "failed", e)
PY
