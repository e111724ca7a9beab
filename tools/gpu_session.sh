#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python tools/bench_decode_batch.py 2>&1 | grep -v amdgpu
