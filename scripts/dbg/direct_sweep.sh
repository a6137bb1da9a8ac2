# resident-query filter, LDS ring (0) against direct row loads (1), measurement build: filter ms of 6 launches, alternating runs
for D in ${DIMS:-256 384 512 640 768}; do for Q in ${QS:-1024 512 256}; do for rep in 1 2 3; do
for DR in 0 1; do echo -n "dim $D q $Q direct $DR: "; DIM=$D Q=$Q YAMS_ACCEL_I8R_DIRECT=$DR PYTHONPATH=. timeout 150 python scripts/dbg/filter_forms.py 2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['2']['filter_ms'],3))"; done; done; done; done
