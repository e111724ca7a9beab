# rocprofv3 kernel stats of the bench command (run on the GPU box through gpurun):  bash tools/prof_bench.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01_x}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8 > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:26]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {float(r["Percentage"]):6.2f}% n={int(r["Calls"]):6d} avg={float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:110]}')
PY
