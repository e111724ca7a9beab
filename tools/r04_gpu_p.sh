#!/bin/bash
# GPU call P: non-temporal K / V loads in every decode attention kernel (default build) against the -DOMCHAT_KV_NT=0 twin (ab_lib/lib_kvnt0.so)
mkdir -p gpurun_out/r04_p
for v in nt1 nt0; do
  if [ $v = nt0 ]; then export OMCHAT_LIB=$PWD/ab_lib/lib_kvnt0.so; else unset OMCHAT_LIB; fi
  echo "== $v" >> gpurun_out/r04_p/bench_attn.log
  python tools/bench_attn_decode.py 32 3700 2>&1 | grep -v amdgpu.ids >> gpurun_out/r04_p/bench_attn.log
  python bench.py --workload configs1 --steps 3 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_p/c1_$v.json 2> gpurun_out/r04_p/c1_$v.err
  python bench.py --workload configs2 --steps 2 --warmup 1 --no-cpu-baseline --no-side > gpurun_out/r04_p/c2_$v.json 2> gpurun_out/r04_p/c2_$v.err
  python bench.py --workload configs3 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_p/c3_$v.json 2> gpurun_out/r04_p/c3_$v.err
  python bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > gpurun_out/r04_p/c4_$v.json 2> gpurun_out/r04_p/c4_$v.err
done
cat gpurun_out/r04_p/bench_attn.log
python - <<'PY'
import json
for c in ("c1", "c2", "c3", "c4"):
    for v in ("nt1", "nt0"):
        try:
            d = json.load(open(f"gpurun_out/r04_p/{c}_{v}.json"))
            print(c, v, "value %.1f" % d["value"], "decode ms %.4f" % d.get("decode_ms_per_token_p50", -1), "ttft", d.get("ttft_ms_p50"))
        except Exception as e:
            print(c, v, "failed", e)
PY
