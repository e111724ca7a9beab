# round 3, GPU call Z: eager stream vs captured hipGraph for the six-launch decode step
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_z
mkdir -p $O
for m in "" "--graph" "" "--graph"; do python3 bench.py --workload configs1 --steps 2 --warmup 1 --gen 256 --no-cpu-baseline --no-fp8 $m > $O/b.json 2>> $O/bench.err; python3 - <<PY
import json; d=json.load(open("$O/b.json"))
print("mode '$m'  value", round(d["value"],1), "decode ms/token", round(d["decode_ms_per_token_p50"],4), "graph", d.get("decode_graph"))
PY
done
