# round 3, GPU call E: persistent 256x256 GEMM: bit-exactness tests, A/B in the bench (tuning key 13), GEMM probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_e
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -k "gemm or fp8" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
python3 tools/bench_gemm_k.py 2 > $O/gemm_k_t2_persist.txt 2>&1; head -6 $O/gemm_k_t2_persist.txt
for k in 1 0 1 0; do python3 bench.py --steps 2 --warmup 1 --gen 64 --no-cpu-baseline --no-side --no-fp8 --tuning 13=$k > $O/bench_p$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_p$k.json")); c=d["configs2"]
print("persist $k  c1 vit/pre", round(d["vit_ms_p50"],2), round(d["prefill_ms_p50"],2), " c2 vit/pre", round(c["vit_ms_p50"],1), round(c["prefill_ms_p50"],1), "fc1", round(d["roofline_vit"]["avg_launch_us"],1), round(c["roofline_vit"]["avg_launch_us"],1), "gateup", round(d["roofline_prefill"]["avg_launch_us"],1), round(c["roofline_prefill"]["avg_launch_us"],1))
PY
done
