# round 3, GPU call W: split-KV merge with 65..1024 partials per head in one or two memory round trips (attn_merge_mid_kernel): parity + configs3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_w
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_round3.py tests/test_gpu_fp8.py -q -k "attn_decode or configs3 or full_size or long or 16k or eng8b" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
python3 bench.py --workload configs3 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline > $O/bench_configs3.json 2> $O/bench.err; python3 bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > $O/bench_configs4.json 2>> $O/bench.err; python3 -c "import json; d=json.load(open(\"$O/bench_configs4.json\")); print(\"configs4 value\", round(d[\"value\"],2), \"decode ms/token\", d.get(\"decode_ms_per_token_p50\"))"; python3 - <<PY
import json; d=json.load(open("$O/bench_configs3.json"))
print("configs3 value", round(d["value"],1), "decode ms/token", round(d.get("decode_ms_per_token_p50", d.get("decode_ms_per_step_p50", 0)),4), "hbm", round(d["decode_hbm_frac"],4))
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs3 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs3.csv; rm -rf $O/stats
grep -E "attn_decode|attn_merge" $O/kernel_stats_configs3.csv | cut -c1-150
