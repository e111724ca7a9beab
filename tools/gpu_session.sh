#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "attn or mha" 2>&1 | tail -2
for rep in 1 2 3; do
for v in new new2; do
  echo "--- $v (rep $rep)"; OMCHAT_LIB=$PWD/ab_lib/$v.so timeout 300 python tools/bench_attn.py 20 1 vit,vit24,dec,dec_b4,long16k 2>&1 | grep -v amdgpu | tr '\n' ' '; echo
done
done
