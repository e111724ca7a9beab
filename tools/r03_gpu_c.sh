# round 3, GPU call C: GEMM fixed-cost probe, KLDS decode attention A/B, new tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_c
mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_tp_single.py tests/test_gpu_round2.py -x -q > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
python3 tools/bench_gemm_k.py 2 > $O/gemm_k_t2.txt 2>&1; cat $O/gemm_k_t2.txt
python3 tools/bench_gemm_k.py 9 > $O/gemm_k_t9.txt 2>&1; head -9 $O/gemm_k_t9.txt
for k in 1 0 1 0; do python3 bench.py --workload configs2 --steps 1 --warmup 1 --gen 64 --no-cpu-baseline --tuning 12=$k > $O/bench_c2_klds$k.json 2>> $O/bench_c2.err; python3 - <<PY
import json; d=json.load(open("$O/bench_c2_klds$k.json")); print("klds $k decode ms/step", d["decode_ms_per_token_p50"], "hbm", d["decode_hbm_frac"])
PY
done
