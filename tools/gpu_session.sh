#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round2.py -q -x -k "tiles_per_wave" 2>&1 | tail -8
