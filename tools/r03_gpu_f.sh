# round 3, GPU call F: persistent GEMM with the original XCD tile order (A/B key 13), decode L2 prefetch on a second stream (A/B key 14)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_f
mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_gpu_round3.py -x -q -k "gemm" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for k in 1 0 1 0; do python3 bench.py --steps 2 --warmup 1 --gen 64 --no-cpu-baseline --no-side --no-fp8 --tuning 13=$k > $O/bench_p$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_p$k.json")); c=d["configs2"]
print("persist $k  c1 vit/pre", round(d["vit_ms_p50"],2), round(d["prefill_ms_p50"],2), " c2 vit/pre", round(c["vit_ms_p50"],1), round(c["prefill_ms_p50"],1), "fc1", round(d["roofline_vit"]["avg_launch_us"],1), round(c["roofline_vit"]["avg_launch_us"],1), "gateup", round(d["roofline_prefill"]["avg_launch_us"],1), round(c["roofline_prefill"]["avg_launch_us"],1))
PY
done
for k in 1 0 1 0; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --no-fp8 --tuning 14=$k > $O/bench_pf$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_pf$k.json"))
print("prefetch $k  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "hbm", round(d["decode_hbm_frac"],4), "gateup us", round(d["roofline"]["avg_launch_us"],1))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats_pf -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --tuning 14=1 > $O/stats_pf.json 2> $O/stats_pf.err
cp $(find $O/stats_pf -name "*kernel_stats.csv" | head -1) $O/kernel_stats_pf.csv; rm -rf $O/stats_pf
head -14 $O/kernel_stats_pf.csv | cut -c1-160
