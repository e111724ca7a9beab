#!/bin/bash
# round 5 measurement: what would removing launches buy a TP = N rank's batch-1 decode step if its two exchanges per layer were free?
# rank 0's shard widths (exchanges no-ops, bench.py --shard-of N) through (a) the tensor-parallel structure (8 launches per layer),
# (b) the one-GPU six-launch structure, (c) the one-launch layer of round 4.  Needs the experiments twin:
#   python -m omchat_amd.build --twin ab_lib/experiments -DOMCHAT_EXPERIMENTS=1
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1 OMCHAT_LIB=$PWD/ab_lib/experiments/libomchat_hip.so
N=${1:-8}
for t in "35=0" "35=1" "35=1,23=1"; do
  python3 bench.py --shard-of $N --workload configs1 --steps 2 --warmup 1 --no-cpu-baseline --no-side --no-fp8 --tuning $t 2> /tmp/err.txt | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('TP=$N rank, tuning $t: decode ms/token', round(d['decode_ms_per_token_p50'],4))
except Exception as e:
    print('tuning $t failed', e); print(open('/tmp/err.txt').read()[-600:])"
done
