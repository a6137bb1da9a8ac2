#!/bin/bash
# round 6: where does the BASELINE config 2 step (1M x 384, Q = 256) spend its time?  lane sweep, gate on / off, kernel trace
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_c2}
mkdir -p $O
python bench.py --only-config2 --config2-lane-sweep 1,2,3,4 --oracle-queries 256 > $O/sweep_gate.json 2> $O/sweep_gate.err
YAMS_ACCEL_MEASURE_LIB=1 YAMS_ACCEL_I8R_NO_FLIP=1 python bench.py --only-config2 --config2-lane-sweep 1,2 --oracle-queries 0 > $O/sweep_noflip.json 2> $O/sweep_noflip.err
YAMS_ACCEL_MEASURE_LIB=1 python bench.py --only-config2 --config2-lane-sweep 1,2 --oracle-queries 0 > $O/sweep_flip_measure.json 2> $O/sweep_flip_measure.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof1 -o c2 --output-format csv -- python $OLDPWD/bench.py --only-config2 --config2-lane-sweep 1 --config2-batches 40 --oracle-queries 0 > $OLDPWD/$O/prof1.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof2 -o c2 --output-format csv -- python $OLDPWD/bench.py --only-config2 --config2-lane-sweep 2 --config2-batches 40 --oracle-queries 0 > $OLDPWD/$O/prof2.log 2>&1)
for p in prof1 prof2; do f=$(find $O/$p -name '*kernel_trace.csv' | head -1); if [ -n "$f" ]; then head -1 $f > $O/$p.trace.csv; tail -3000 $f >> $O/$p.trace.csv; rm -f $f; fi; done
cat $O/sweep_gate.json; cat $O/sweep_noflip.json; cat $O/sweep_flip_measure.json
