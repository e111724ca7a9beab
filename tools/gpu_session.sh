#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 1500 bash tools/pmc_gemm.sh > gpurun_out/pmc_gemm_summary.txt 2>&1; echo "pmc_gemm rc=$?"
timeout 1500 bash tools/pmc_attn.sh 1 pmc_attn_r02b > gpurun_out/pmc_attn_summary.txt 2>&1; echo "pmc_attn rc=$?"
tail -30 gpurun_out/pmc_gemm_summary.txt
tail -40 gpurun_out/pmc_attn_summary.txt
