#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ulimit -c 0
REPO=$PWD; OUT=$REPO/gpurun_out
export YAMS_ACCEL_SHA_SLOTS=${YAMS_ACCEL_SHA_SLOTS:-1}
B="python $REPO/scripts/ingest_bench.py --gib ${GIB:-100} --reps 1"
rm -rf $OUT/prof_ing1 $OUT/prof_ing2 $OUT/prof_ing3 $OUT/prof_ingest
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ingest -o ingest -- $B > $OUT/prof_ingest.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_ing1 -o ing -- $B > $OUT/prof_ing1.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -f csv -d $OUT/prof_ing2 -o ing -- $B > $OUT/prof_ing2.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/prof_ing3 -o ing -- $B > $OUT/prof_ing3.log 2>&1) || true
