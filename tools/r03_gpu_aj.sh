# round 3, GPU call AJ: fp8-KV decode, RoPE launch with the position by value: parity + configs4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_aj
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py -q -k "fp8_kv or rope or kv" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for i in 1 2; do python3 bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > $O/b4.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b4.json')); print('configs4 value', round(d['value'],2), 'decode ms/token', round(d['decode_ms_per_token_p50'],4), 'ttft', round(d['ttft_ms_p50'],1))"; done
