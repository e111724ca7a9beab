#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_v
timeout 1700 python3 tools/dbg_tp8.py 8 f16 3 > gpurun_out/r04_v/dbg_tp8_f16.log 2>&1; echo rc=$?
grep -v amdgpu.ids gpurun_out/r04_v/dbg_tp8_f16.log | tail -30
