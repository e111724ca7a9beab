# SQ counters of the prefill attention kernels (run on the GPU box):  bash tools/pmc_attn.sh <gen: 1 | 0> <tag> [shapes of tools/bench_attn.py, default vit24,dec_b4]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
GEN=${1:-1}
TAG=${2:-pmc_attn_g$GEN}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/$TAG/p$i -o r --output-format csv -- python3 $R/tools/bench_attn.py 3 $GEN ${3:-vit24,dec_b4} > $R/gpurun_out/${TAG}_$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(R+"/gpurun_out/$TAG/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:48], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,d in agg.items():
    print("kernel/grid", g)
    for k,v in sorted(d.items()): print(f"  {k:36s} {sum(v)/len(v):16.0f}")
    m={k:sum(v)/len(v) for k,v in d.items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:      # GUI summed over 8 XCDs, MFMA busy over 1024 SIMDs
        print(f"  -> MFMA pipe busy = {m['SQ_VALU_MFMA_BUSY_CYCLES']/(m['GRBM_GUI_ACTIVE']*128):.3f} of the kernel's cycles")
PY
