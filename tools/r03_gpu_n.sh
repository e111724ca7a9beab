# round 3, GPU call N: long-K kernel with one row per wave (three register buffers): tests + A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_n
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_model.py tests/test_gpu_fp8.py tests/test_gpu_ops.py tests/test_gpu_graph.py tests/test_gpu_api.py -q -k "norm or decode or gemv or fp8 or graph or generate or forward" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for k in 3 1 3 1; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --tuning 14=$k > $O/bench_n$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_n$k.json"))
print("key14=$k  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "hbm", round(d["decode_hbm_frac"],4), "gateup us", round(d["roofline"]["avg_launch_us"],1), "fp8 ms/token", round(d["fp8_decode"]["decode_ms_per_token"],4))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1.csv; rm -rf $O/stats
grep -E "gemv_rows|resid_rmsnorm|attn_decode|attn_merge|rmsnorm" $O/kernel_stats_configs1.csv | cut -c1-150
