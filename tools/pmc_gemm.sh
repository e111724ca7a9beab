# SQ counters of the GEMM kernels on the hot-path shapes (run on the GPU box):  bash tools/pmc_gemm.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_gemm/p$i -o r --output-format csv -- python3 $R/tools/bench_gemm.py 2 0.05 > $R/gpurun_out/pmc_gemm_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/pmc_gemm/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm8_kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][-40:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,d in sorted(agg.items()):
    print(g)
    v={k:sum(x)/len(x) for k,x in d.items()}
    for k in sorted(v): print(f"  {k:30s} {v[k]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: busy fraction = busy / (GUI / 8 * 1024)
        print(f"  -> MFMA pipe busy = {v['SQ_VALU_MFMA_BUSY_CYCLES']/(v['GRBM_GUI_ACTIVE']*128):.3f} of the kernel's cycles ({v['GRBM_GUI_ACTIVE']/8/1e3:.0f} k cycles per XCD)")
PY
