#!/bin/bash
# GPU call U2: TP = 8 same-engine check, tiny geometry first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_u
run() { name=$1; shift; OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 900 python3 bench.py "$@" --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r04_u/$name.json 2> gpurun_out/r04_u/$name.err; echo "$name rc=$?"; python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04_u/$name.json")); c = d.get("tp1_check") or {}
    print("   match", d.get("tokens_match_tp1"), "rel_err %.4g" % c.get("logit_rel_err", -1), "equal", c.get("equal"), "/", c.get("compared"), "check dtype", c.get("dtype"), "ttft", d.get("ttft_ms_p50"))
except Exception as e:
    print("   no line:", e)
PY
tail -2 gpurun_out/r04_u/$name.err | cut -c1-300; }
run tiny8_f16_same --gpus 8 --tiny --dtype f16
run tiny8_bf16_same --gpus 8 --tiny --dtype bf16 --tp1-check-dtype bf16
run tiny8_bf16_f16check --gpus 8 --tiny --dtype bf16
run tiny4_f16_same --gpus 4 --tiny --dtype f16
run full4_f16_same --gpus 4 --dtype f16 --workload configs1
run full8_f16_c1 --gpus 8 --dtype f16 --workload configs1
