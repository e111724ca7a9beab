#!/bin/bash
# round 6, GPU call D: the whole GPU suite on the fused ViT layer + clean A/B of tuning key 44 (no profiler), configs1 and configs2 (24-tile launches)
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
mkdir -p gpurun_out/r06_d
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 > gpurun_out/r06_d/pytest_gpu.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/r06_d/pytest_gpu.txt
cp gpurun_out/fulldepth_parity.json gpurun_out/r06_d/ 2>/dev/null
for i in 1 2; do
bash tools/gpu_job.sh r06_d ab 44 0 1 --workload configs1 --steps 5 --warmup 2 --gen 8
done
for v in 0 1 1 0; do
python3 bench.py --workload configs2 --steps 2 --warmup 1 --gen 4 --no-cpu-baseline --no-side --no-fp8 --tuning 44=$v > gpurun_out/r06_d/c2_$v.json 2> gpurun_out/r06_d/c2_$v.err
python3 -c "
import json; d=json.load(open('gpurun_out/r06_d/c2_$v.json')); print('configs2 key 44 = $v:', {k: round(d[k],3) for k in ('vit_ms_p50','prefill_ms_p50','vit_mfma_frac') if k in d})"
done
