#!/bin/bash
# GPU call AB: whole GPU suite + smoke on the current code
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_ab
python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r04_ab/pytest_gpu.txt
cat gpurun_out/r04_ab/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
