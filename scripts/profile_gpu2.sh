#!/bin/bash
# PMC passes for the scan filter kernel (each counter group in its own run).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
B="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest"
rm -rf $OUT/prof_trace $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3 $OUT/prof_pmc4
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_trace -o scan -- $B > $OUT/prof_trace.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_pmc1 -o scan -- $B > $OUT/prof_pmc1.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/prof_pmc2 -o scan -- $B > $OUT/prof_pmc2.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --kernel-trace -f csv -d $OUT/prof_pmc3 -o scan -- $B > $OUT/prof_pmc3.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -f csv -d $OUT/prof_pmc4 -o scan -- $B > $OUT/prof_pmc4.log 2>&1) || true
tail -3 $OUT/prof_pmc3.log $OUT/prof_pmc4.log | cut -c1-200
