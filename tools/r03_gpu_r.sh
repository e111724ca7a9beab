# round 3, GPU call R: balanced o_proj (7 waves) and qkv (9 waves) workgroups, tuning key 17 = 1 / 0: parity + A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_r
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_round3.py -q -k "norm or six_launch or long_k or balanced or gemv" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for k in 1 0 1 0; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --tuning 17=$k > $O/bench_n$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_n$k.json"))
print("key17=$k  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "hbm", round(d["decode_hbm_frac"],4), "gateup us", round(d["roofline"]["avg_launch_us"],1), "fp8 ms/token", round(d["fp8_decode"]["decode_ms_per_token"],4))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --tuning 17=1 > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1_key17_1.csv; rm -rf $O/stats
grep -E "gemv_rows|attn_decode|attn_merge" $O/kernel_stats_configs1_key17_1.csv | cut -c1-150
