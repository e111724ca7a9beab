# round 3, GPU call D: spill-free epilogue with hoisted loads: GEMM probe, tune new unchunked TP classes, op tests, bench, shard8
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_d
mkdir -p $O
cd $R
python3 tools/bench_gemm_k.py 2 > $O/gemm_k_t2.txt 2>&1; head -9 $O/gemm_k_t2.txt
python3 tools/gen_gemm_tune.py > $O/tune.log 2>&1; tail -2 $O/tune.log
cp omchat_amd/gemm_tune_gfx950.txt $O/gemm_tune_gfx950.txt
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_round3.py tests/test_gpu_fp8.py tests/test_gpu_tp_single.py -x -q > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $O/bench_n1.json 2> $O/bench_n1.err; head -c 1300 $O/bench_n1.json; echo
python3 bench.py --shard-of 8 --steps 2 --warmup 1 > $O/bench_shard8.json 2> $O/bench_shard8.err
python3 bench.py --shard-of 2 --steps 2 --warmup 1 > $O/bench_shard2.json 2> $O/bench_shard2.err
python3 bench.py --shard-of 4 --steps 2 --warmup 1 > $O/bench_shard4.json 2> $O/bench_shard4.err
python3 - <<PY
import json
for n in (2,4,8):
    d=json.load(open("$O/bench_shard%d.json"%n)); c=d["configs2"]
    print(n, "c1 vit/pre/dec", round(d["vit_ms_p50"],1), round(d["prefill_ms_p50"],1), round(d["decode_ms_per_token_p50"],3), "c2", round(c["vit_ms_p50"],1), round(c["prefill_ms_p50"],1), round(c["decode_ms_per_step_p50"],3))
PY
