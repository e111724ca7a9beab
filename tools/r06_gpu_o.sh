#!/bin/bash
# round 6, call O: per-kernel times (rocprofv3 kernel stats) of the e4m3-cache decode attention forms at one context length
cd /tmp && export TMPDIR=/tmp
export OMCHAT_ALLOW_TUNING=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_o}; mkdir -p $O; cd $R
for L in ${2:-33300}; do
  rm -rf $O/p
  rocprofv3 --kernel-trace --stats -d $O/p -o p --output-format csv -- python3 tools/bench_attn_decode_kv8.py ${3:-1} $L > $O/bench_$L.txt 2> $O/bench_$L.err
  f=$(find $O/p -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/kernel_stats_kv8_attn_$L.csv
  rm -rf $O/p
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/kernel_stats_kv8_attn_$L.csv")):
    n = r["Name"]
    if "attn" in n: print("L $L  %-110s calls %5s avg %8.2f us" % (n[:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
