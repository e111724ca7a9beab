#!/bin/bash
# round 6, call T: kernel stats of a configs[4] step per tuning setting (decode kernels only printed)
cd /tmp && export TMPDIR=/tmp
export OMCHAT_ALLOW_TUNING=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_t}; mkdir -p $O; cd $R
shift
for t in "$@"; do
  rm -rf $O/p
  rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- python3 bench.py --workload configs4 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side --tuning $t > $O/bench_$t.json 2> $O/bench_$t.err
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_configs4_$t.csv
  rm -rf $O/p
  python3 - <<PY
import csv
tot = 0
for r in csv.DictReader(open("$O/kernel_stats_configs4_$t.csv")):
    n = r["Name"]
    if any(k in n for k in ("attn_decode", "attn_merge", "rope_kv", "gemv")): print("$t  %-100s calls %5s avg %8.2f us" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
