# round 4, GPU call D: launch-latency environment knobs on the configs1 decode loop (A/B in separate processes on one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_d
run() { # name, env...
  n=$1; shift
  env "$@" timeout 600 python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8 --tuning 22=0 > gpurun_out/r04_d/$n.json 2> gpurun_out/r04_d/$n.err
  python3 - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_d/$n.json").read().strip().splitlines()[-1])
    print("$n", "value", round(d["value"], 1), "decode ms", round(d["decode_ms_per_token_p50"], 4), "ttft", round(d["ttft_ms_p50"], 2), "generate", round(d.get("generate_tokens_per_sec", 0), 1))
except Exception as e:
    print("$n failed", e); print(open("gpurun_out/r04_d/$n.err").read()[-800:])
PY
}
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run base2 A=1
run hwq1 GPU_MAX_HW_QUEUES=1
run devkernarg1_hwq2 HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=2
