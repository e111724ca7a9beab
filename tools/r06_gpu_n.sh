#!/bin/bash
# round 6, call N: the e4m3-cache decode attention with a wave walking 2 .. 4 tiles (tuning key 47): op-level parity, kernel-level A/B at the
# configs[3] / configs[4] context lengths, then the model-level A/B on configs[4]
cd /tmp && export TMPDIR=/tmp
export OMCHAT_ALLOW_TUNING=1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r06_n}; mkdir -p $O; cd $R
timeout 600 python3 -m pytest tests/test_gpu_fp8.py -q -x -k "kv8_decode_attention or fp8_kv_cache" 2>&1 | tail -5 | tee $O/pytest_kv8.txt
for L in 3700 8800 16500 33300; do timeout 300 python3 tools/bench_attn_decode_kv8.py 1 $L; done 2>&1 | grep "^kv8" | tee $O/kv8_attn_ab.txt
timeout 300 python3 tools/bench_attn_decode_kv8.py 4 8800 2>&1 | grep "^kv8" | tee -a $O/kv8_attn_ab.txt
for v in 1 0; do
  timeout 900 python3 bench.py --workload configs4 --steps 2 --warmup 1 --no-cpu-baseline --no-side --tuning 47=$v > $O/configs4_key47_$v.json 2> $O/configs4_key47_$v.err
  python3 - <<PY
import json
d = json.load(open("$O/configs4_key47_$v.json"))
print("key 47 = $v:", {k: round(d[k], 4) for k in ("value", "decode_ms_per_token_p50", "decode_hbm_frac", "prefill_ms_p50") if d.get(k)})
PY
done 2>&1 | tee $O/configs4_ab.txt
