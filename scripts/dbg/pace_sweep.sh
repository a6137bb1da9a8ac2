#!/bin/bash
# Pacing interval of the resident-query filter: launch time and HBM bytes per launch (FETCH_SIZE x 2, gfx950) for
# "look at the siblings' counters every 2^n-th strip", dims 768 and 384, 1024 queries, 12.5M rows (measurement build).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
OUT=$PWD/gpurun_out/pace; mkdir -p $OUT; : > $OUT/pace.txt
for D in ${DIMS:-768 384}; do for P in ${PACES:-0 2 3 4 5 31}; do
  T=$(DIM=$D Q=1024 YAMS_ACCEL_I8R_PACE_LOG2=$P timeout 200 python scripts/dbg/filter_forms.py 80 2>/dev/null | tail -1)
  rm -rf $OUT/pmc_${D}_$P
  (cd /tmp && DIM=$D Q=1024 YAMS_ACCEL_I8R_PACE_LOG2=$P timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/pmc_${D}_$P -o q -- python $OLDPWD/scripts/dbg/filter_forms.py 80 > /dev/null 2>&1) || true
  F=$(python - <<PY
import csv,glob
v=[]
for f in glob.glob("$OUT/pmc_${D}_$P/**/q_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_tiles_i8r_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": v.append(float(r["Counter_Value"]))
print(round(sum(v)/len(v)*2*1024/1e9,2) if v else None, len(v))
PY
)
  echo "dim $D pace_log2 $P: $T hbm_GB_per_launch(x2) $F" | tee -a $OUT/pace.txt
  rm -rf $OUT/pmc_${D}_$P
done; done
