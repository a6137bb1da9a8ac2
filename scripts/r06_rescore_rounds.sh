#!/bin/bash
# round 6: rounds of 128 candidates (product) against 256 / 192 / one round (measurement build knobs) — headline, config 2, non-uniform legs
F="--steps 20 --warmup 5 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"
for rep in 1 2; do for form in 0 2 4; do
  YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_RESCORE_FORM=$form python bench.py $F 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head form $form', d['ms_per_step'], d['roofline']['launch_ms'])"
done; done
for form in 0 2 4; do
  YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_RESCORE_FORM=$form python bench.py --only-config2 --config2-lane-sweep 2 --oracle-queries 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 form $form', d['ms_per_step'], d['launch_ms'])"
  YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_RESCORE_FORM=$form python bench.py --only-distribution --oracle-queries 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist form $form', [(k, round(v['ms_per_step'],2)) for k,v in d.items()])"
done
