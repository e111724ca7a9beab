# round 3, GPU call T: fresh SQ counter passes of the prefill attention and of the GEMMs (closing code of round 3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_t
bash tools/pmc_attn.sh 1 r03_t/pmc_attn > gpurun_out/r03_t/pmc_attn.txt 2>&1
bash tools/pmc_gemm.sh > gpurun_out/r03_t/pmc_gemm.txt 2>&1
rm -rf gpurun_out/r03_t/pmc_attn gpurun_out/pmc_gemm
tail -30 gpurun_out/r03_t/pmc_attn.txt; tail -12 gpurun_out/r03_t/pmc_gemm.txt
