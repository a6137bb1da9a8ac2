#!/bin/bash
# round 6: the product-quantised engine's bench leg + its kernel stats
export TMPDIR=/tmp
REPO=$PWD; O=$REPO/gpurun_out/r06_pq; mkdir -p $O
python bench.py --only-pq > $O/pq.json 2> $O/pq.err; tail -c 1800 $O/pq.json; echo
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o pq -- python $REPO/bench.py --only-pq --oracle-queries 0 > $O/prof.log 2>&1) || true
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r06_pq/prof/**/*kernel_stats.csv', recursive=True)+glob.glob('gpurun_out/r06_pq/prof/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
    break
PY
