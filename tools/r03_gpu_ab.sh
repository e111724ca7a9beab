# round 3, GPU call AB: position bookkeeping folded into the argmax's second stage: parity subset + bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ab
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_graph.py tests/test_gpu_api.py tests/test_gpu_round2.py tests/test_gpu_tp_single.py -q -k "decode or generate or forward or graph or greedy or argmax or free_running or tp" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for i in 1 2; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --no-fp8 > $O/b.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b.json')); print('value', round(d['value'],1), 'decode ms/token', round(d['decode_ms_per_token_p50'],4))"; done
