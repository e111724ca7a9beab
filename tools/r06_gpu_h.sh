#!/bin/bash
# round 6, GPU call H: closing evidence on the current tree -- the driver's sequence (pytest -m gpu, smoke, default bench line), the roofline table with the
# PMC passes (configs1 + configs2), kernel stats + PMC of configs3 / configs4, the live oracle pass that pins the fixture and times the CPU baseline
cd $GRAFT_REPO_ROOT
export OMCHAT_ALLOW_TUNING=1
T=${1:-r06_h}
mkdir -p gpurun_out/$T
bash tools/gpu_job.sh $T suite
cp gpurun_out/fulldepth_parity.json gpurun_out/$T/ 2>/dev/null; cp gpurun_out/fp8_per_layer.json gpurun_out/$T/ 2>/dev/null
bash tools/gpu_job.sh $T table 2>&1 | tail -30
bash tools/gpu_job.sh $T table34 2>&1 | tail -5
OMCHAT_LIVE_ORACLE=1 timeout 900 python3 -m pytest tests/test_gpu_fulldepth.py -q -k live -s 2>&1 | tail -5
cp gpurun_out/oracle_cpu_phases.json gpurun_out/$T/ 2>/dev/null
