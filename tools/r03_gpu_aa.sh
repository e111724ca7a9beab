# round 3, GPU call AA: full -m gpu suite + the per-round evidence pass (tools/collect_profiles.sh r03_aa)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_aa
timeout 2400 python3 -m pytest tests -q -m gpu > gpurun_out/r03_aa/pytest_gpu.log 2>&1; tail -3 gpurun_out/r03_aa/pytest_gpu.log
bash tools/collect_profiles.sh r03_aa
