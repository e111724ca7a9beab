#!/bin/bash
# round 6, GPU call A: new tests (dead-row skip bit-identity, fixture-based full-depth parity), in-model A/B of tuning key 43, configs[3]/[4] evidence
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_a
timeout 900 python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_fulldepth.py -q -x --durations=10 2>&1 | tail -40 > gpurun_out/r06_a/pytest_new.txt; tail -15 gpurun_out/r06_a/pytest_new.txt
cp gpurun_out/fulldepth_parity.json gpurun_out/r06_a/ 2>/dev/null
bash tools/gpu_job.sh r06_a ab 43 0 1 --workload configs1 --steps 3 --warmup 1
bash tools/gpu_job.sh r06_a ab 43 1 0 --workload configs1 --steps 3 --warmup 1
bash tools/gpu_job.sh r06_a table34 2>&1 | tail -60
