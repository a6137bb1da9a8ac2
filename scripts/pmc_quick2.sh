#!/bin/bash
# wait/LDS counters of the filter kernel and its ablations; output under gpurun_out/pmcq2
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
OUT=$PWD/gpurun_out; rm -rf $OUT/pmcq2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -f csv -d $OUT/pmcq2 -o q -- python $OLDPWD/scripts/filter_ablation.py "$@" > $OUT/pmcq2.log 2>&1) || true
python - <<'PY'
import csv, os, collections
G=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out","pmcq2")
dur={}
for r in csv.DictReader(open(os.path.join(G,"q_kernel_trace.csv"))):
    dur[r["Dispatch_Id"]]=(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
vals=collections.defaultdict(dict)
for r in csv.DictReader(open(os.path.join(G,"q_counter_collection.csv"))):
    vals[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
for d,(n,ns) in dur.items():
    if "scan_tiles_bf16" in n and "<1" in n:
        v=vals.get(d,{})
        wc=v.get("SQ_WAVE_CYCLES",1)
        print(n[40:75], "ms=%.2f"%(ns/1e6), {k: round(x/wc,3) for k,x in v.items() if k not in ("SQ_WAVE_CYCLES",)}, "wave_cycles=%.3g"%wc)
PY
