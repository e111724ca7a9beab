#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python tools/ab_decode2.py 0,8,9,10,12,13,14,15,11 eager 2>&1 | grep "skip="
