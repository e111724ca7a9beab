#!/bin/bash
# round 6, call S: configs[4] decode with the e4m3-cache attention forms, same box: one wave per tile (47=1), walking form with the RoPE + append
# launch in front (48=0), walking form with it folded in (defaults)
cd /tmp && export TMPDIR=/tmp
export OMCHAT_ALLOW_TUNING=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_s}; mkdir -p $O; cd $R
for t in 47=1 48=0 48=1 47=1 48=1; do
  timeout 900 python3 bench.py --workload configs4 --steps 2 --warmup 1 --no-cpu-baseline --no-side --tuning $t > $O/configs4_$t.json 2> $O/configs4_$t.err
  python3 - <<PY
import json
d = json.load(open("$O/configs4_$t.json"))
print("tuning $t:", {k: round(d[k], 4) for k in ("value", "decode_ms_per_token_p50", "decode_hbm_frac", "prefill_ms_p50") if d.get(k)})
PY
done 2>&1 | tee $O/configs4_ab.txt
