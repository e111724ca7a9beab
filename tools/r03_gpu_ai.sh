# round 3, GPU call AI: full -m gpu suite + the per-round evidence pass (tools/collect_profiles.sh r03_ai)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_ai
timeout 2400 python3 -m pytest tests -q -m gpu > gpurun_out/r03_ai/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03_ai/pytest_gpu.log
bash tools/collect_profiles.sh r03_ai
