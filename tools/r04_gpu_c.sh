# round 4, GPU call C: timeline of the fused attention + o_proj launch (diagnostic build with stamps)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_c
timeout 300 tools/bin/tune_fused 3648 > gpurun_out/r04_c/fused_timeline.txt 2>&1; echo rc=$?
cat gpurun_out/r04_c/fused_timeline.txt
