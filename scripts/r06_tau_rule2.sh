#!/bin/bash
# round 6: the proof-aware threshold, second pass — config 2 kernel stats (cost of tau_select_kernel), config 2 three times, the legs, scan tests
export TMPDIR=/tmp
REPO=$PWD; O=$REPO/gpurun_out/r06_tau2; mkdir -p $O
C2="python $REPO/bench.py --only-config2 --config2-lane-sweep 2 --config2-batches 200 --oracle-queries 0"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_c2 -o c2 -- $C2 > $O/prof_c2.log 2>&1) || true
grep -h "tau_select\|topk_block\|rescore_select\|scan_tiles" $O/prof_c2/*kernel_stats.csv $O/prof_c2/*/*kernel_stats.csv 2>/dev/null | cut -c1-60,120-260
for i in 1 2 3; do python bench.py --only-config2 > $O/c2_$i.json 2> $O/c2_$i.err; python - $O/c2_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('config2', d.get('ms_per_step'), d.get('qps'), d.get('launch_ms'), d.get('results_identical_to_the_oracle_run'))
PY
done
python bench.py --only-distribution > $O/dist.json 2> $O/dist.err; tail -c 1900 $O/dist.json
python -m pytest tests -q -m gpu > $O/pytest_scan.txt 2>&1; grep -n "passed\|failed" $O/pytest_scan.txt | tail -3
