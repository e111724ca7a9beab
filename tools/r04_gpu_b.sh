# round 4, GPU call B: the round-4 tests (fused attention + o_proj launch, padded-batch decode), the decode / generate tests that the
# changed code touches, then the configs1 bench with the fused launch on and off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_b
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/r04_b/pytest_r4.log 2>&1; echo "round4 tests rc=$?"; tail -25 gpurun_out/r04_b/pytest_r4.log
timeout 900 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_gpu_model.py -x -q > gpurun_out/r04_b/pytest_api.log 2>&1; echo "api tests rc=$?"; tail -5 gpurun_out/r04_b/pytest_api.log
timeout 600 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 > gpurun_out/r04_b/bench_fused.json 2> gpurun_out/r04_b/bench_fused.err; echo "bench fused rc=$?"
timeout 600 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 --tuning 22=0 > gpurun_out/r04_b/bench_unfused.json 2> gpurun_out/r04_b/bench_unfused.err; echo "bench unfused rc=$?"
python3 - <<'PY'
import json
for n in ("fused", "unfused"):
    try:
        d = json.loads(open(f"gpurun_out/r04_b/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, "value", round(d["value"], 1), "decode ms", round(d["decode_ms_per_token_p50"], 4), "hbm", round(d["decode_hbm_frac"], 4), "ttft", round(d["ttft_ms_p50"], 2),
              "generate", d.get("generate"), "fused", d.get("fused_decode"))
    except Exception as e:
        print(n, "failed", e); print(open(f"gpurun_out/r04_b/bench_{n}.err").read()[-1500:])
PY
