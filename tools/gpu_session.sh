#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round2.py -q -x -k "gemv_packed" 2>&1 | tail -3
