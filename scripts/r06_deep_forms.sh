#!/bin/bash
# round 6: scan_tiles_i8d_kernel (two slabs of row fragments in flight per wave; measurement version 90) against the shipped direct
# form (87): candidate sets on the ragged / masked shapes first, then launch times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/${1:-r06_deep}
mkdir -p $O
YAMS_ACCEL_MEASURE_LIB=1 FORMS_VERSIONS=2,87,90 timeout 600 python tests/_filter_forms.py > $O/identical.json 2> $O/identical.err
tail -c 1200 $O/identical.json; tail -3 $O/identical.err
for rep in 1 2; do
  ROWS=12500000 DIM=768 Q=1024 timeout 300 python scripts/dbg/filter_forms.py 87 90 > $O/d768_q1024_$rep.json 2> /dev/null; cat $O/d768_q1024_$rep.json; echo
  ROWS=12500000 DIM=384 Q=1024 timeout 300 python scripts/dbg/filter_forms.py 87 90 > $O/d384_q1024_$rep.json 2> /dev/null; cat $O/d384_q1024_$rep.json; echo
done
ROWS=1000000 DIM=384 Q=256 timeout 300 python scripts/dbg/filter_forms.py 87 90 87 90 > $O/c2.json 2> /dev/null; cat $O/c2.json; echo
ROWS=12500000 DIM=384 Q=256 timeout 300 python scripts/dbg/filter_forms.py 87 90 > $O/d384_q256.json 2> /dev/null; cat $O/d384_q256.json; echo
ROWS=12500000 DIM=768 Q=256 timeout 300 python scripts/dbg/filter_forms.py 87 90 > $O/d768_q256.json 2> /dev/null; cat $O/d768_q256.json; echo
ROWS=12500000 DIM=768 Q=64 timeout 300 python scripts/dbg/filter_forms.py 87 90 > $O/d768_q64.json 2> /dev/null; cat $O/d768_q64.json; echo
