#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ak; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/p -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > /dev/null 2>&1
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/p
python3 - <<PY
import csv
for r in list(csv.DictReader(open("$O/kernel_stats.csv")))[:24]:
    print(f"  {r['Name'][:96]:96s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
