#!/bin/bash
# GPU call W: 8 rank processes on one GPU, f16 main run (the check then runs on the measured engine itself), peer grids of 8 workgroups
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_w
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 1500 python3 bench.py --gpus 8 --dtype f16 --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r04_w/bench_os_8_f16.json 2> gpurun_out/r04_w/bench_os_8_f16.err; echo rc=$?
python3 - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r04_w/bench_os_8_f16.json")); print({k: d[k] for k in ("tokens_match_tp1", "tp1_check", "peer_timeouts", "comm_stats", "ttft_ms_p50", "decode_ms_per_token_p50")})
except Exception as e:
    print("no line", e)
PY
grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r04_w/bench_os_8_f16.err | tail -5 | cut -c1-400
