# round 4, GPU call G: one-launch layer: round-4 tests (bit identity in three modes), timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_g
timeout 900 python3 -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/r04_g/pytest_r4.log 2>&1; echo "round4 tests rc=$?"; tail -6 gpurun_out/r04_g/pytest_r4.log
timeout 300 tools/bin/tune_layer 3648 > gpurun_out/r04_g/layer_timeline.txt 2>&1; echo "timeline rc=$?"; cat gpurun_out/r04_g/layer_timeline.txt | head -24
