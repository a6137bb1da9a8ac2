# A/B of the resident-query filter with direct row loads (YAMS_ACCEL_I8R_DIRECT, measurement build) through bench.py:
# alternating runs, cosine headline + the L2 leg
for i in 1 2 3; do for DR in 0 1; do
YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_I8R_DIRECT=$DR python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-ingest --no-hbm-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --oracle-queries 0 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('direct', $DR, 'step', round(d['ms_per_step'],3), 'filter', round(d['roofline']['launch_ms'],3), 'L2 step', round(d['config3_l2']['ms_per_step'],3), 'L2 filter', round(d['config3_l2']['launch_ms'],3), 'L2 q64', round(d['config3_l2']['q64']['ms_per_step'],3))"
done; done
