#!/bin/bash
# round 6, GPU call G: row statistics by LDS-DMA -- op tests, ViT goldens, full-depth fixture, A/B of tuning key 44, roofline table
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_g
timeout 1200 python3 -m pytest tests/test_gpu_round6.py tests/test_gpu_fulldepth.py tests/test_gpu_model.py tests/test_gpu_round2.py tests/test_gpu_fullsize.py -q -x --durations=5 2>&1 | tail -12
bash tools/gpu_job.sh r06_g ab 44 0 1 --workload configs1 --steps 5 --warmup 2 --gen 8
bash tools/gpu_job.sh r06_g ab 44 1 0 --workload configs1 --steps 5 --warmup 2 --gen 8
bash tools/gpu_job.sh r06_g table 2>&1 | grep -E "ViT|unlabelled|prefill"
