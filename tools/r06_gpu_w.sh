#!/bin/bash
# round 6, GPU call W: the closing sequence of r06_gpu_h.sh on the final tree, then the tests that skip in the product build against the experiments
# twin (built beforehand in the container: python -m omchat_amd.build --twin ab_lib/experiments -DOMCHAT_EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT
T=${1:-r06_w}
bash tools/r06_gpu_h.sh $T
export OMCHAT_ALLOW_TUNING=1
if [ -f ab_lib/experiments/libomchat_hip.so ]; then
  OMCHAT_LIB=$PWD/ab_lib/experiments/libomchat_hip.so timeout 1200 python3 -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -q 2>&1 | tail -4 | tee gpurun_out/$T/pytest_experiments_twin.txt
fi
