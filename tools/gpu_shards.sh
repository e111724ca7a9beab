#!/bin/bash
# per-rank shard lines (inputs of tools/tp_projection.py) + the N = 1 line of the same box:  tools/gpu_shards.sh <tag>
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side > $O/bench_n1.json 2> $O/bench_n1.err
for n in 2 4 8; do python3 bench.py --shard-of $n --steps 2 --warmup 1 > $O/bench_shard$n.json 2> $O/bench_shard$n.err; done
python3 tools/tp_projection.py $O/bench_n1.json $O/bench_shard2.json $O/bench_shard4.json $O/bench_shard8.json | tee $O/tp_projection.txt
