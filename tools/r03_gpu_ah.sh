# round 3, GPU call AH: e4m3 weights loaded 16 bytes per lane (norm GEMV + long-K kernels): parity + fp8 decode block + configs4
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_ah
mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_fp8.py tests/test_gpu_round3.py tests/test_gpu_tp_single.py -q -k "fp8 or norm or long_k or six_launch" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -3
for i in 1 2; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline > $O/b1.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b1.json')); print('configs1 value', round(d['value'],1), 'decode', round(d['decode_ms_per_token_p50'],4), 'fp8 decode ms/token', round(d['fp8_decode']['decode_ms_per_token'],4))"; done
python3 bench.py --workload configs4 --steps 2 --warmup 1 --gen 64 --no-cpu-baseline > $O/b4.json 2>> $O/bench.err; python3 -c "
import json; d=json.load(open('$O/b4.json')); print('configs4 value', round(d['value'],2), 'decode ms/token', round(d['decode_ms_per_token_p50'],4))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload configs1 --steps 1 --warmup 1 --gen 64 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats.json 2> $GRAFT_REPO_ROOT/$O/stats.err
cd $GRAFT_REPO_ROOT
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1_fp8.csv; rm -rf $O/stats
grep -E "Lb1|true" $O/kernel_stats_configs1_fp8.csv | cut -c1-170
