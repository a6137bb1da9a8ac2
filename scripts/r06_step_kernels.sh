#!/bin/bash
# round 6: the step's kernels one lane at a time (durations without the other lane's sweep in front of them)
export TMPDIR=/tmp
REPO=$PWD; O=$REPO/gpurun_out/r06_step1; mkdir -p $O
B="python $REPO/bench.py --steps 20 --warmup 5 --lanes 1 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o scan -- $B > $O/prof.log 2>&1) || true
python - <<'PY'
import csv,glob
f=(glob.glob('gpurun_out/r06_step1/prof/*kernel_stats.csv')+glob.glob('gpurun_out/r06_step1/prof/**/*kernel_stats.csv', recursive=True))[0]
tot=0
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=20:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>4s} {float(r['AverageNs'])/1e3:9.1f} us"); tot+=float(r['AverageNs'])/1e3*(int(r['Calls'])/26.0)
print("sum per step (us):", round(tot,1))
PY
$B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one lane: ms_per_step', d['ms_per_step'], 'launch', d['roofline']['launch_ms'])"
