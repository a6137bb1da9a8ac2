#!/bin/bash
# round 6: workgroups per query of i8_collect_sample_kernel (48 at the bench shape) — single lane, kernel stats
export TMPDIR=/tmp
REPO=$PWD; O=$REPO/gpurun_out/r06_gx; mkdir -p $O
B="python $REPO/bench.py --steps 20 --warmup 5 --lanes 1 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"
for gx in 64 16 8 4 2; do
  rm -rf $O/prof
  (cd /tmp && YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_COLLECT_GX=$gx timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o scan -- $B > $O/prof.log 2>&1) || true
  python - $gx <<'PY'
import csv,glob,sys
f=(glob.glob('gpurun_out/r06_gx/prof/*kernel_stats.csv')+glob.glob('gpurun_out/r06_gx/prof/**/*kernel_stats.csv', recursive=True))[0]
for r in csv.DictReader(open(f)):
    if 'collect_sample' in r['Name']: print('gx', sys.argv[1], 'collect', round(float(r['AverageNs'])/1e3,1), 'us')
PY
done
