# round 4, GPU call I: dynamic gate|up (key 24): bit identity + configs1 bench A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_i
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q -k "dynamic_gate_up" > gpurun_out/r04_i/pytest.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04_i/pytest.log
for k in 1 0; do
  timeout 600 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 --tuning 24=$k > gpurun_out/r04_i/bench_k$k.json 2> gpurun_out/r04_i/bench_k$k.err; echo "bench key24=$k rc=$?"
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/r04_i/bench_k$k.json").read().strip().splitlines()[-1])
print("key24=$k value", round(d["value"], 1), "decode ms", round(d["decode_ms_per_token_p50"], 4), "hbm", round(d["decode_hbm_frac"], 4), "roofline us", round(d["roofline"]["avg_launch_us"], 2), d["roofline"]["kernel"][:40])
PY
done
