#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "attn or mha" > gpurun_out/m_tests1.log 2>&1; echo "attn tests rc=$?"
tail -5 gpurun_out/m_tests1.log
echo "--- v3 (pipelined)"; timeout 300 python tools/bench_attn.py 20 2
echo "--- v2"; timeout 300 python tools/bench_attn.py 20 1
echo "--- v1"; timeout 300 python tools/bench_attn.py 20 0
echo "--- v3 again"; timeout 300 python tools/bench_attn.py 20 2
