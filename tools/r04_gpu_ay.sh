#!/bin/bash
# GPU call AY: regression of the N-rank bench path on the closing code (2 and 4 rank processes on ONE GPU, peer transport; timings meaningless)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ay
for n in 2 4; do
  OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 900 python3 bench.py --gpus $n --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r04_ay/bench_os_$n.json 2> gpurun_out/r04_ay/bench_os_$n.err; echo "n=$n rc=$?"
  python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04_ay/bench_os_$n.json")); c = d.get("tp1_check") or {}
    print("   match", d.get("tokens_match_tp1"), "rel_err %.4g" % c.get("logit_rel_err", -1), "equal", c.get("equal"), "/", c.get("compared"), "guarded", c.get("guarded_equal"), "/", c.get("guarded"), "timeouts", d.get("peer_timeouts"), "vit dp", (d.get("vit_data_parallel") or {}).get("features_equal_tp"))
except Exception as e:
    print("   no line:", e)
PY
  grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r04_ay/bench_os_$n.err | tail -2 | cut -c1-300
done
