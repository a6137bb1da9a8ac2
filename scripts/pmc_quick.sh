#!/bin/bash
# quick PMC look at the filter kernel and its ablations (clock, MFMA busy); output under gpurun_out/pmcq
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
OUT=$PWD/gpurun_out; rm -rf $OUT/pmcq
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -f csv -d $OUT/pmcq -o q -- python $OLDPWD/scripts/filter_ablation.py "$@" > $OUT/pmcq.log 2>&1) || true
python - <<'PY'
import csv, os, collections
G=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out","pmcq")
dur={}
for r in csv.DictReader(open(os.path.join(G,"q_kernel_trace.csv"))):
    dur[r["Dispatch_Id"]]=(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
vals=collections.defaultdict(dict)
for r in csv.DictReader(open(os.path.join(G,"q_counter_collection.csv"))):
    vals[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
for d,(n,ns) in dur.items():
    if "scan_tiles_bf16" in n and "ILi1" in n or ("scan_tiles_bf16" in n and "<1" in n):
        v=vals.get(d,{})
        gui=v.get("GRBM_GUI_ACTIVE",0)/8.0
        print(n[:70], "ms=%.2f"%(ns/1e6), "clk_GHz=%.2f"%(gui/ns if ns else 0), "mfma_busy=%.3f"%(v.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/(gui*1024) if gui else 0))
PY
