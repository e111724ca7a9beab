#!/bin/bash
# round 6, GPU call B: the fused ViT layer -- op tests, model goldens, full-depth fixture, in-model A/B of tuning key 44
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_b
timeout 1200 python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_fulldepth.py tests/test_gpu_model.py tests/test_gpu_round2.py -q -x --durations=5 2>&1 | tail -40 > gpurun_out/r06_b/pytest.txt; tail -25 gpurun_out/r06_b/pytest.txt
cp gpurun_out/fulldepth_parity.json gpurun_out/r06_b/ 2>/dev/null
bash tools/gpu_job.sh r06_b ab 44 0 1 --workload configs1 --steps 3 --warmup 1
bash tools/gpu_job.sh r06_b ab 44 1 0 --workload configs1 --steps 3 --warmup 1
