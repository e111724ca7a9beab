# round 3, GPU call AE: split-KV merge with column groups (tuning key 21): parity + configs4 / configs3 A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ae
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -q -k "attn_decode" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for k in 1 0 1 0; do python3 bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline --tuning 21=$k > $O/b4.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b4.json')); print('configs4 key21=$k value', round(d['value'],2), 'decode ms/token', round(d['decode_ms_per_token_p50'],4))"; done
for k in 1 2 1 2; do python3 bench.py --workload configs3 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --tuning 21=$k > $O/b3.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b3.json')); print('configs3 key21=$k value', round(d['value'],2), 'decode ms/token', round(d['decode_ms_per_token_p50'],4))"; done
