#!/bin/bash
# GPU call Z2: TTFT with the heaviest-first GQA prefill attention; configs1 / configs2 / configs3 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_z
python bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_z/c1.json 2> gpurun_out/r04_z/c1.err
python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_z/c2.json 2> gpurun_out/r04_z/c2.err
python bench.py --workload configs3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_z/c3.json 2> gpurun_out/r04_z/c3.err
python - <<'PY'
import json
for c in ("c1", "c2", "c3"):
    try:
        d = json.load(open(f"gpurun_out/r04_z/{c}.json")); print(c, "value %.1f" % d["value"], "decode ms %.4f" % d.get("decode_ms_per_token_p50", -1), "ttft %.2f vit %.2f prefill %.2f" % (d.get("ttft_ms_p50", -1), d.get("vit_ms_p50", -1), d.get("prefill_ms_p50", -1)), "prefill frac", d.get("prefill_mfma_frac"))
    except Exception as e: print(c, "failed", e)
PY
