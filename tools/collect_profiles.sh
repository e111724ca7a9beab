# Per-round evidence (run on the GPU box through gpurun):  bash tools/collect_profiles.sh <tag>
# kernel stats of a WARMED step (rocprofv3 --kernel-trace --stats; --warmup 1 and the committed tile-choice file keep first-use GEMM
# tuning out of the trace), HBM traffic (two separate --pmc passes), then the bench line with the CPU baseline.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03_x}
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats -o s --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline --no-fp8 > $O/stats.json 2> $O/stats.err
rocprofv3 --kernel-trace --stats -d $O/stats2 -o s --output-format csv -- python3 $R/bench.py --workload configs2 --steps 1 --warmup 1 --gen 16 --no-cpu-baseline --no-fp8 > $O/stats2.json 2> $O/stats2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --workload configs1 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 > $O/write.json 2> $O/write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch2 -o f --output-format csv -- python3 $R/bench.py --workload configs2 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 > $O/fetch2.json 2> $O/fetch2.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write2 -o w --output-format csv -- python3 $R/bench.py --workload configs2 --steps 1 --warmup 0 --gen 4 --no-cpu-baseline --no-fp8 > $O/write2.json 2> $O/write2.err
rocprofv3 --kernel-trace --stats -d $O/stats3 -o s --output-format csv -- python3 $R/bench.py --workload configs3 --steps 1 --warmup 1 --gen 32 --no-cpu-baseline > $O/stats3.json 2> $O/stats3.err
cd $R
python3 tools/pmc_summary.py $(find $O/fetch -name "*counter_collection.csv" | head -1) $(find $O/write -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
python3 tools/pmc_summary.py $(find $O/fetch2 -name "*counter_collection.csv" | head -1) $(find $O/write2 -name "*counter_collection.csv" | head -1) $O/pmc_traffic_configs2.json > $O/pmc_summary_configs2.txt 2>&1
cp $(find $O/stats3 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs3.csv
cp $O/pmc_traffic_configs2.json $R/profiles/${TAG}_pmc_traffic_configs2.json
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs1.csv
cp $(find $O/stats2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_configs2.csv
cp $O/pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json
# the bench line below reads the files profiles/MANIFEST.json names (bench.py::pmc_traffic), never "the last one in sort order"
python3 - <<PY
import json
json.dump({"note": "which committed PMC pass bench.py::pmc_traffic reads (written by tools/collect_profiles.sh with the files themselves)",
           "pmc_traffic": "${TAG}_pmc_traffic.json", "pmc_traffic_configs2": "${TAG}_pmc_traffic_configs2.json"}, open("$R/profiles/MANIFEST.json", "w"), indent=1)
PY
python3 bench.py > $O/bench.json 2> $O/bench.err
rm -rf $O/stats $O/stats2 $O/stats3 $O/fetch $O/write $O/fetch2 $O/write2      # keep the folded summaries, drop the raw traces (gpurun_out is size-capped)
head -c 600 $O/bench.json; echo; head -12 $O/pmc_summary.txt; head -8 $O/pmc_summary_configs2.txt
