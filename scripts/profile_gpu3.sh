#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD; OUT=$REPO/gpurun_out
B="python $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ingest"
rm -rf $OUT/prof_pmc3 $OUT/prof_pmc2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -f csv -d $OUT/prof_pmc3 -o scan -- $B > $OUT/prof_pmc3.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT --kernel-trace -f csv -d $OUT/prof_pmc2 -o scan -- $B > $OUT/prof_pmc2.log 2>&1) || true
