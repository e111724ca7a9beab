# round 3, GPU call O: prefill attention, K operand reads ahead of the S^T MFMAs (+ 3-deep ring variant): A/B and parity
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_o
mkdir -p $O
cd $R
python3 tools/bench_attn_ab.py 15 12,2,3 20 > $O/attn_ab.txt 2>&1; cat $O/attn_ab.txt
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_round2.py tests/test_gpu_round3.py -q -k "attn or attention" > $O/pytest.log 2>&1; grep -E "passed|failed|Error" $O/pytest.log | tail -5
