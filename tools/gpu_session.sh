#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for v in base xmask base xmask; do
echo "--- $v"; OMCHAT_LIB=$PWD/ab_lib/$v.so timeout 900 python tools/bench_decode_batch.py 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
done
timeout 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py -q -x -k "packed or gemv" 2>&1 | tail -2
