cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
REPO=$PWD; OUT=$REPO/gpurun_out
B="python $REPO/bench.py --no-cpu-baseline --no-ingest --no-hbm-leg"
rm -rf $OUT/prof_trace $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_trace -o scan -- $B > $OUT/prof_trace.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_pmc1 -o scan -- $B > $OUT/prof_pmc1.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace -f csv -d $OUT/prof_pmc2 -o scan -- $B > $OUT/prof_pmc2.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum --kernel-trace -f csv -d $OUT/prof_pmc3 -o scan -- $B > $OUT/prof_pmc3.log 2>&1) || true
tail -2 $OUT/prof_trace.log | cut -c1-300
