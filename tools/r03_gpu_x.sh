# round 3, GPU call X: the 512-thread merge form below 64 partials per head (tuning key 19 = 64 default / 16): A/B at the configs1 context
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_x
mkdir -p $O
cd $R
for k in 64 16 64 16; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --no-fp8 --tuning 19=$k > $O/bench_n$k.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/bench_n$k.json"))
print("key19=$k  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "hbm", round(d["decode_hbm_frac"],4))
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --tuning 19=16 > $O/stats.json 2> $O/stats.err
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1_key19_16.csv; rm -rf $O/stats
grep -E "attn_decode|attn_merge" $O/kernel_stats_configs1_key19_16.csv | cut -c1-150
