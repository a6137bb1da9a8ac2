#!/bin/bash
# Several PMC passes (one counter group per run) over scripts/filter_ablation.py "$@"; prints, per filter
# launch, every counter and its ratio to the kernel's cycles.  Output under gpurun_out/pmcg<i>.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
OUT=$PWD/gpurun_out
GROUPS_=(
 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
 "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
 "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE"
 "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE"
 "FETCH_SIZE GRBM_GUI_ACTIVE"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
)
if [ -n "$PMC_ONLY" ]; then GROUPS_=("${GROUPS_[$PMC_ONLY]}"); fi
i=0
for g in "${GROUPS_[@]}"; do
  rm -rf $OUT/pmcg$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $g --kernel-trace -f csv -d $OUT/pmcg$i -o q -- python $OLDPWD/scripts/filter_ablation.py "$@" > $OUT/pmcg$i.log 2>&1) || true
  i=$((i+1))
done
python - <<'PY'
import csv, os, collections, glob
R=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out")
for G in sorted(glob.glob(os.path.join(R,"pmcg[0-9]"))):
    try:
        dur={}
        for r in csv.DictReader(open(os.path.join(G,"q_kernel_trace.csv"))):
            dur[r["Dispatch_Id"]]=(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
        vals=collections.defaultdict(dict)
        for r in csv.DictReader(open(os.path.join(G,"q_counter_collection.csv"))):
            vals[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
    except Exception as e:
        print(G, "failed", e); continue
    for d,(n,ns) in dur.items():
        if ("scan_tiles_i8" in n and "<1" in n) or "scan_tiles_i8r" in n:
            v=vals.get(d,{})
            cyc=v.get("GRBM_GUI_ACTIVE",0)/8.0
            print(os.path.basename(G), n[17:50], "ms=%.2f cyc=%.3g"%(ns/1e6,cyc), {k: "%.4g (%.3f/cyc)"%(x, x/cyc if cyc else 0) for k,x in v.items() if k!="GRBM_GUI_ACTIVE"})
PY
