cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
REPO=$PWD; OUT=$REPO/gpurun_out
B="python $REPO/bench.py --no-cpu-baseline --no-ingest --no-hbm-leg --lanes 1 --oracle-queries 0"
rm -rf $OUT/prof_lane1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_lane1 -o scan -- $B > $OUT/prof_lane1.log 2>&1) || true
grep -o '"ms_per_step": [0-9.]*' $OUT/prof_lane1.log
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/prof_lane1/scan_kernel_stats.csv"))))
for r in rows[:18]:
    if int(r['Calls'])>=10: print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f}")
PY
