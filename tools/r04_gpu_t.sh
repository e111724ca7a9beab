#!/bin/bash
# GPU call T: the N-rank bench path with 8 (and 4) rank PROCESSES sharing the one GPU (peer transport over hipIpc, no RCCL; timings meaningless):
# TP = 8 logits against the TP = 1 context in f16 and bf16, guarded ids
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_t
for dt in f16 bf16; do
  for n in 8; do
    OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 1200 python3 bench.py --gpus $n --dtype $dt --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r04_t/bench_os_${n}_$dt.json 2> gpurun_out/r04_t/bench_os_${n}_$dt.err
    echo "n=$n dt=$dt rc=$?"; head -c 1200 gpurun_out/r04_t/bench_os_${n}_$dt.json; echo; tail -3 gpurun_out/r04_t/bench_os_${n}_$dt.err
  done
done
