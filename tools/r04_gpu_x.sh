#!/bin/bash
# GPU call X: fp32 partial sums under tensor parallelism (tuning key 29): tests, then the 8-rank (one GPU) f16 logit error against TP = 1 with it on
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_x
python -m pytest tests/test_gpu_tp_single.py tests/test_gpu_peer.py -q -x 2>&1 | tail -5 > gpurun_out/r04_x/test.log; cat gpurun_out/r04_x/test.log
for k in 1 0; do
OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 1500 python3 bench.py --gpus 8 --dtype f16 --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side --tuning 29=$k > gpurun_out/r04_x/bench_os_8_f16_f32sum$k.json 2> gpurun_out/r04_x/bench_os_8_f16_f32sum$k.err; echo rc=$?
python3 - <<PY
import json
try:
    d = json.load(open("gpurun_out/r04_x/bench_os_8_f16_f32sum$k.json")); print("key29=$k", {k: d[k] for k in ("tokens_match_tp1", "tp1_check", "peer_timeouts")})
except Exception as e:
    print("no line", e)
PY
grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/r04_x/bench_os_8_f16_f32sum$k.err | tail -3 | cut -c1-300
done
