#!/bin/bash
# GPU call Q: slots x stages of the LDS-DMA decode attention inside the configs2 decode step; configs4 with the fp8 rows back on plain loads
mkdir -p gpurun_out/r04_q
for k in 1026 516 515 771 ; do
  python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side --tuning 26=$k > gpurun_out/r04_q/c2_$k.json 2> gpurun_out/r04_q/c2_$k.err
done
python bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > gpurun_out/r04_q/c4.json 2> gpurun_out/r04_q/c4.err
python tools/bench_attn_decode.py 32 3700 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_q/bench_attn.log
cat gpurun_out/r04_q/bench_attn.log
python - <<'PY'
import json
for c in ("c2_1026", "c2_516", "c2_515", "c2_771", "c4"):
    try:
        d = json.load(open(f"gpurun_out/r04_q/{c}.json"))
        print(c, "value %.1f" % d["value"], "decode ms %.4f" % d.get("decode_ms_per_token_p50", -1))
    except Exception as e:
        print(c, "failed", e)
PY
