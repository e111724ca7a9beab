# round 3, GPU call U: five-launch decode layer (tuning key 18 = 3 / 1 / 0): parity + A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_u
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_fp8.py -q -k "decode or merge or gemv or generate or forward or graph" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5
for k in 3 1 0 3 1 0; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --tuning 18=$k > $O/bench_n$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_n$k.json"))
print("key18=$k  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "hbm", round(d["decode_hbm_frac"],4), "fp8 ms/token", round(d["fp8_decode"]["decode_ms_per_token"],4))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1.csv; rm -rf $O/stats
grep -E "gemv_rows|attn_decode|attn_merge" $O/kernel_stats_configs1.csv | cut -c1-150
