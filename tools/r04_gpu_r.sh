#!/bin/bash
# GPU call R: where does the LDS-DMA form start to pay?  b = 4 / 8 / 16 at 3.7 k keys, one-tile kernel (auto) against the forced ring form; long contexts at b = 1
mkdir -p gpurun_out/r04_r
for b in 4 8 12; do
  python tools/bench_attn_decode.py $b 3700 0 0 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r04_r/bench_attn.log
  python tools/bench_attn_decode.py $b 3700 0 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_r/bench_attn.log
done
python tools/bench_attn_decode.py 1 33280 0 0 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r04_r/bench_attn.log
python tools/bench_attn_decode.py 1 33280 0 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_r/bench_attn.log
python tools/bench_attn_decode.py 4 33280 0 0 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r04_r/bench_attn.log
python tools/bench_attn_decode.py 4 33280 0 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_r/bench_attn.log
python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side --tuning 26=514 > gpurun_out/r04_r/c2_514.json 2> gpurun_out/r04_r/c2_514.err
cat gpurun_out/r04_r/bench_attn.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_r/c2_514.json")); print("c2 2x2", "value %.1f" % d["value"], "decode ms %.4f" % d.get("decode_ms_per_token_p50", -1))
PY
