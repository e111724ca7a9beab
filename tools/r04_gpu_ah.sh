#!/bin/bash
# GPU call AH: the closing code of round 3 (ab_lib/r03_tree) against the current tree on one box: configs1 line + warmed kernel stats of the same command
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_ah; mkdir -p $O
for v in r04 r03 r04 r03; do
  if [ $v = r03 ]; then D=$R/ab_lib/r03_tree; else D=$R; fi
  (cd $D && python3 bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'value %.1f decode %.4f ttft %.2f vit %.2f prefill %.2f' % (d['value'], d['decode_ms_per_token_p50'], d['ttft_ms_p50'], d['vit_ms_p50'], d['prefill_ms_p50']))")
done
for v in r04 r03; do
  if [ $v = r03 ]; then D=$R/ab_lib/r03_tree; else D=$R; fi
  rocprofv3 --kernel-trace --stats -d $O/p_$v -o s --output-format csv -- python3 $D/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 --no-side > /dev/null 2>&1
  cp $(find $O/p_$v -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$v.csv; rm -rf $O/p_$v
done
python3 - <<PY
import csv
def load(f):
    return {r['Name']: (int(r['Calls']), float(r['AverageNs']) / 1e3) for r in csv.DictReader(open(f))}
a, b = load("$O/kernel_stats_r04.csv"), load("$O/kernel_stats_r03.csv")
import re
def key(n): return re.sub(r'Lb0ELb0E', 'Lb0E', n)[:90]
bb = {key(k): v for k, v in b.items()}
for k, (c, t) in sorted(a.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:22]:
    o = bb.get(key(k))
    print(f"{k[:80]:80s} r04 {c:5d} x {t:8.1f} us   r03 " + (f"{o[0]:5d} x {o[1]:8.1f} us" if o else "-"))
PY
