#!/bin/bash
# round 6: the proof-aware threshold — stage trace of the clustered leg (measurement build), the two non-uniform legs as the
# bench runs them, the headline and config 2 for regressions
O=gpurun_out/r06_tau; mkdir -p $O
YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_TRACE_STAGES=1 python bench.py --only-distribution --distribution clustered --lanes 1 --steps 4 --warmup 2 --oracle-queries 0 > $O/clu_trace.json 2> $O/clu_trace.err
grep "^stage" $O/clu_trace.err | tail -12
python bench.py --only-distribution > $O/dist.json 2> $O/dist.err; tail -c 1800 $O/dist.json
python bench.py --only-config2 > $O/c2.json 2> $O/c2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_tau/c2.json').read().strip().splitlines()[-1]); print('config2', d.get('ms_per_step'), d.get('qps'), d.get('results_identical_to_the_oracle_run'))
PY
python bench.py --steps 20 --warmup 5 --no-ingest --no-cpu-baseline --no-config2-leg --no-distribution-legs --no-c-abi-leg --no-boundary-leg --no-l2-leg --no-hbm-leg --oracle-queries 64 > $O/head.json 2> $O/head.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_tau/head.json').read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['value'], d['roofline']['launch_ms'], d['roofline']['frac'], d.get('bit_exact_vs_oracle'))
PY
