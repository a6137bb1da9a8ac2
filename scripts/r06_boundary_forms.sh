#!/bin/bash
# round 6: the strip boundary of scan_tiles_i8r_kernel — 85 = shipped (ZSM 6), 86 = accumulators born from row block 0, 87 = boundary work
# before the drain, 88 = both.  Candidate sets first (tests/_filter_forms.py), then launch times on the bench shard and at dim 384.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r06_forms}
mkdir -p $O
YAMS_ACCEL_MEASURE_LIB=1 FORMS_VERSIONS=2,85,86,87,88 python tests/_filter_forms.py > $O/identical.json 2> $O/identical.err
tail -c 1500 $O/identical.json
for rep in 1 2; do
  ROWS=12500000 DIM=768 Q=1024 python scripts/dbg/filter_forms.py 85 86 87 88 > $O/d768_q1024_$rep.json 2> /dev/null; cat $O/d768_q1024_$rep.json; echo
  ROWS=12500000 DIM=384 Q=1024 python scripts/dbg/filter_forms.py 85 86 87 88 > $O/d384_q1024_$rep.json 2> /dev/null; cat $O/d384_q1024_$rep.json; echo
done
ROWS=1000000 DIM=384 Q=256 python scripts/dbg/filter_forms.py 85 86 87 88 85 88 > $O/c2.json 2> /dev/null; cat $O/c2.json; echo
ROWS=12500000 DIM=768 Q=64 python scripts/dbg/filter_forms.py 85 86 87 88 > $O/d768_q64.json 2> /dev/null; cat $O/d768_q64.json; echo
