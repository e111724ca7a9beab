#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -q -x -k "packed or gemv or batch or decode" 2>&1 | tail -3
