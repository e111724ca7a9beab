# round 4, GPU call J: the whole -m gpu suite on the current code
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_j
timeout 3000 python3 -m pytest tests -q -m gpu > gpurun_out/r04_j/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -15 gpurun_out/r04_j/pytest_gpu.log
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -q -s -k configs4 2>&1 | grep -a "configs4 whole model" | tail -2
