# round 3, GPU call AC: regression after the device_cus() refactor (launch-shape selection must be unchanged): GEMV / decode tests + bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ac
mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_fp8.py -q -k "gemv or decode or norm or balanced or packed" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-side > $O/b.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b.json')); print('value', round(d['value'],1), 'decode ms/token', round(d['decode_ms_per_token_p50'],4), 'configs2 decode', d['configs2']['decode_ms_per_step_p50'], 'tune', d['gemm_tune_measurements'])"
