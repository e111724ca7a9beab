#!/bin/bash
# GPU call L: kernel-level A/B of the batched decode attention forms (+ rocprofv3 kernel stats of the same tool)
mkdir -p gpurun_out/r04_l
python tools/bench_attn_decode.py 32 3700 > gpurun_out/r04_l/bench_attn.log 2>&1
python tools/bench_attn_decode.py 16 3700 >> gpurun_out/r04_l/bench_attn.log 2>&1
python tools/bench_attn_decode.py 32 1024 >> gpurun_out/r04_l/bench_attn.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_l/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_attn_decode.py 32 3700 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/r04_l/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/r04_l/kernel_stats.csv && rm -rf gpurun_out/r04_l/prof
cat gpurun_out/r04_l/bench_attn.log; head -8 gpurun_out/r04_l/kernel_stats.csv | cut -c1-200
