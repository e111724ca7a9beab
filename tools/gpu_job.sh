#!/bin/bash
# One parameterised GPU job script (replaces the per-call tools/r0N_gpu_*.sh of earlier rounds):  tools/gpu_job.sh <tag> <job> [args...]
# Outputs land in gpurun_out/<tag>/.  Jobs:
#   shardstats N        rocprofv3 kernel stats of rank 0's share of a TP = N group (configs1 gen 32, configs2 gen 16)
#   bench [args]        bench.py with the given arguments -> bench.json
#   stats [args]        rocprofv3 --kernel-trace --stats of bench.py with the given arguments
#   pytest [args]       python -m pytest with the given arguments
#   py script [args]    python3 script args
#   suite               the driver's round-end sequence: pytest -m gpu (durations recorded), smoke(), bench.py default line
#   ab key v1 v2 [args] bench.py once per value of tuning key `key` (same box), decode / ViT / prefill figures side by side
#   table34             configs[3] and configs[4]: kernel stats of a warmed step + FETCH_SIZE / WRITE_SIZE passes -> kernel_stats_configs{3,4}_*.csv, pmc_traffic_configs{3,4}.json
#   table               configs[1]: kernel trace + FETCH_SIZE / WRITE_SIZE passes -> roofline_table.txt (tools/roofline_table.py), kernel stats CSV
cd /tmp && export TMPDIR=/tmp
export OMCHAT_ALLOW_TUNING=1      # tools/*.py set tuning keys (process-global measurement hooks)
R=$GRAFT_REPO_ROOT
tag=$1; job=$2; shift 2
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
stats() {   # name, bench args...
  local name=$1; shift
  rm -rf $O/prof_$name
  rocprofv3 --kernel-trace --stats -d $O/prof_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/${name}.json 2> $O/${name}.err
  f=$(find $O/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_${name}.csv && head -40 $O/kernel_stats_${name}.csv | cut -c1-200
  rm -rf $O/prof_$name
}
case $job in
  shardstats)
    n=$1
    stats shard${n}_configs1_gen32 --shard-of $n --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side --no-fp8
    stats shard${n}_configs2_gen16 --shard-of $n --workload configs2 --steps 1 --warmup 1 --gen 16 --no-cpu-baseline --no-side --no-fp8
    ;;
  bench) python3 bench.py "$@" > $O/bench.json 2> $O/bench.err; head -c 3000 $O/bench.json; tail -3 $O/bench.err ;;
  stats) stats run "$@" ;;
  pytest) python3 -m pytest "$@" 2>&1 | tail -30 | tee $O/pytest.txt ;;
  suite)
    timeout 1500 python3 -m pytest tests -m gpu -q --durations=40 > $O/pytest_gpu.txt 2>&1; echo "gpu tests rc=$?"; tail -3 $O/pytest_gpu.txt
    timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
    timeout 1200 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; head -c 1500 $O/bench_final.json
    ;;
  ab)
    key=$1; shift; vals="$1 $2"; shift 2
    for v in $vals; do
      python3 bench.py --no-cpu-baseline --no-side --no-fp8 --tuning $key=$v "$@" > $O/ab_${key}_$v.json 2> $O/ab_${key}_$v.err
      python3 - <<PY
import json
d = json.load(open("$O/ab_${key}_$v.json")); c = d.get("configs2") or {}
print("key $key = $v:", {k: round(d[k], 4) for k in ("value", "decode_ms_per_token_p50", "vit_ms_p50", "prefill_ms_p50") if d.get(k)}, {k: round(c[k], 4) for k in ("decode_ms_per_step_p50",) if c.get(k)})
PY
    done
    ;;
  table)
    B="--workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --no-side"
    rm -rf $O/t $O/f $O/w
    rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 bench.py $B > $O/table_bench.json 2> $O/table_bench.err
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 bench.py --workload configs1 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2> $O/fetch.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 bench.py --workload configs1 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2> $O/write.err
    T=$(find $O/t -name "*kernel_trace.csv" | head -1); F=$(find $O/f -name "*counter_collection.csv" | head -1); W=$(find $O/w -name "*counter_collection.csv" | head -1)
    cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1_steps1_gen32.csv
    python3 tools/roofline_table.py $T --fetch $F --write $W | tee $O/roofline_table.txt
    python3 tools/pmc_summary.py $F $W $O/pmc_traffic.json > /dev/null
    rm -rf $O/t $O/f $O/w
    # configs[2] (batch 32): kernel stats + the two PMC passes -> pmc_traffic_configs2.json (bench.py reads it for the side block's traffic)
    rocprofv3 --kernel-trace --stats -d $O/t2 -o t --output-format csv -- python3 bench.py --workload configs2 --steps 1 --warmup 1 --gen 16 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2> $O/stats2.err
    cp $(find $O/t2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs2_steps1_gen16.csv
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f2 -o f --output-format csv -- python3 bench.py --workload configs2 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2> $O/fetch2.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w2 -o w --output-format csv -- python3 bench.py --workload configs2 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2> $O/write2.err
    python3 tools/pmc_summary.py $(find $O/f2 -name "*counter_collection.csv" | head -1) $(find $O/w2 -name "*counter_collection.csv" | head -1) $O/pmc_traffic_configs2.json > /dev/null
    rm -rf $O/t2 $O/f2 $O/w2
    ;;
  table34)
    for w in configs3 configs4; do
      g=32; [ $w = configs4 ] && g=32
      B="--workload $w --steps 1 --warmup 1 --gen $g --no-cpu-baseline"
      rm -rf $O/t $O/f $O/w
      rocprofv3 --kernel-trace --stats -d $O/t -o t --output-format csv -- python3 bench.py $B > $O/stats_$w.json 2> $O/stats_$w.err
      cp $(find $O/t -name "*kernel_stats.csv" | head -1) $O/kernel_stats_${w}_steps1_gen$g.csv && head -24 $O/kernel_stats_${w}_steps1_gen$g.csv | cut -c1-180
      rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --gen 4 --no-cpu-baseline > /dev/null 2> $O/fetch_$w.err
      rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 bench.py --workload $w --steps 1 --warmup 0 --gen 4 --no-cpu-baseline > /dev/null 2> $O/write_$w.err
      python3 tools/pmc_summary.py $(find $O/f -name "*counter_collection.csv" | head -1) $(find $O/w -name "*counter_collection.csv" | head -1) $O/pmc_traffic_$w.json | head -14
      rm -rf $O/t $O/f $O/w
    done
    ;;
  py) python3 "$@" 2>&1 | tee $O/py.txt | tail -60 ;;
  *) echo "unknown job $job"; exit 2 ;;
esac
