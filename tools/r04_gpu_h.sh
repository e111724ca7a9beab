cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_h
timeout 300 tools/bin/tune_layer 3648 > gpurun_out/r04_h/layer_timeline.txt 2>&1; echo "timeline rc=$?"; cat gpurun_out/r04_h/layer_timeline.txt
