# round 3, GPU call V: regression of the multi-rank bench path on ONE GPU (ranks share the device: peer transport over hipIpc, no RCCL)
# + smoke() + the TP tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_v
for n in 2 4; do OMCHAT_BENCH_OVERSUBSCRIBE=1 timeout 900 python3 bench.py --gpus $n --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-side > gpurun_out/r03_v/bench_os_$n.json 2> gpurun_out/r03_v/bench_os_$n.err; echo "rc=$?"; head -c 900 gpurun_out/r03_v/bench_os_$n.json; echo; tail -2 gpurun_out/r03_v/bench_os_$n.err; done
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
