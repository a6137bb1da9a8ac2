#!/bin/bash
# round 6: pacing interval / window of the sibling pacing for the deep-prefetch sweep (defaults: pace_log2 2, window 1), headline loop
F="--steps 20 --warmup 5 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"
for rep in 1 2; do for cfg in "2 1" "3 1" "4 1" "1 1" "2 2" "3 2"; do set -- $cfg
  YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_I8R_PACE_LOG2=$1 YAMS_ACCEL_I8R_WINDOW=$2 python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pace_log2 $1 window $2:', d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
