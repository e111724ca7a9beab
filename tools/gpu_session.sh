#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "--- gen 3 (staggered)"; timeout 300 python tools/bench_attn.py 5 3 dec,dec_b16,long16k,long33k
echo "--- gen 1"; timeout 300 python tools/bench_attn.py 5 1 dec,dec_b16,long16k,long33k
