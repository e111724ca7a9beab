# round 4, GPU call E: one-launch decoder layer: bit-identity tests, timeline harness, configs1 bench A/B (key 23 on / off)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_e
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q -k "one_launch or repeatable" > gpurun_out/r04_e/pytest_r4.log 2>&1; echo "round4 tests rc=$?"; tail -25 gpurun_out/r04_e/pytest_r4.log
timeout 300 tools/bin/tune_layer 3648 > gpurun_out/r04_e/layer_timeline.txt 2>&1; echo "timeline rc=$?"; cat gpurun_out/r04_e/layer_timeline.txt
for k in 1 0; do
  timeout 600 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 --tuning 23=$k > gpurun_out/r04_e/bench_k$k.json 2> gpurun_out/r04_e/bench_k$k.err; echo "bench key23=$k rc=$?"
done
python3 - <<'PY'
import json
for n in ("k1", "k0"):
    try:
        d = json.loads(open(f"gpurun_out/r04_e/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "value", round(d["value"], 1), "decode ms", round(d["decode_ms_per_token_p50"], 4), "hbm", round(d["decode_hbm_frac"], 4), "ttft", round(d["ttft_ms_p50"], 2),
              "generate", round(d.get("generate_tokens_per_sec", 0), 1), "fused", d.get("fused_decode", {}).get("launches"), d.get("fused_decode", {}).get("timeout_bits"))
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r04_e/bench_{n}.err").read()[-1500:])
PY
