#!/bin/bash
# round 6: headline step vs launch — three plain runs, then the kernel stats of the same command
export TMPDIR=/tmp
REPO=$PWD; O=$REPO/gpurun_out/r06_head; mkdir -p $O
B="python $REPO/bench.py --steps 20 --warmup 5 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"
for i in 1 2 3; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('run', d['ms_per_step'], d['roofline']['launch_ms'])"; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o scan -- $B > $O/prof.log 2>&1) || true
python - <<'PY'
import csv,glob
f=(glob.glob('gpurun_out/r06_head/prof/*kernel_stats.csv')+glob.glob('gpurun_out/r06_head/prof/**/*kernel_stats.csv', recursive=True))[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>=20: print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
