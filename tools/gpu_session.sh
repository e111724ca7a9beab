#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; R=$PWD
cd /tmp
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_w/p$i -o r --output-format csv -- python3 $R/tools/bench_gemm.py 2,10 > $R/gpurun_out/pmc_w_$i.log 2>&1
  echo "set $i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc_w/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "gemm8" in n and r["Grid_Size"] in ("524288",):     # sq8192: 1024 tiles x 512
            agg["wide" if "gemm8w" in n else "narrow"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,d in sorted(agg.items()):
    print(g)
    v={k:sum(x)/len(x) for k,x in d.items()}
    for k in sorted(v): print(f"  {k:30s} {v[k]:16.0f}")
PY
