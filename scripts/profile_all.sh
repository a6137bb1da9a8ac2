#!/bin/bash
# Round profile recipe: kernel-trace + stats for the bench (scan) and the ingest leg, then PMC passes
# for the dominant scan kernel, each counter group in its own run.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
ulimit -c 0
REPO=$PWD; OUT=$REPO/gpurun_out
B="python $REPO/bench.py --steps 20 --warmup 5 --oracle-queries 0 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --no-config2-leg --no-distribution-legs --no-pq-leg"   # (the HBM leg runs the same kernel at Q = 64: profiled apart, below, so that this trace averages ONE workload)
# (bench.py builds the bf16 filter shadow before the timed region; shadow_build_kernel shows up once in the trace)
I="python $REPO/scripts/ingest_bench.py --gib 100 --reps 2"
rm -rf $OUT/prof_trace $OUT/prof_ingest $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_pmc3
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_trace -o scan -- $B > $OUT/prof_trace.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ingest -o ingest -- $I > $OUT/prof_ingest.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_pmc1 -o scan -- $B > $OUT/prof_pmc1.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VALU --kernel-trace -f csv -d $OUT/prof_pmc2 -o scan -- $B > $OUT/prof_pmc2.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum --kernel-trace -f csv -d $OUT/prof_pmc3 -o scan -- $B > $OUT/prof_pmc3.log 2>&1) || true
# small batches: the narrow (HBM-bound) filter at Q = 64
S="python $REPO/scripts/small_batch.py"
rm -rf $OUT/prof_small $OUT/prof_small_pmc
(cd /tmp && QS=64 FORMS=default timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_small -o small -- $S > $OUT/small_batch_prof.jsonl 2> $OUT/prof_small.log) || true
(cd /tmp && QS=64 FORMS=default timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_small_pmc -o small -- $S > /dev/null 2> $OUT/prof_small_pmc.log) || true
# ingest: VALU issue counters of the SHA-256 / CDC kernels (their roofline is integer VALU throughput), and HBM reads
rm -rf $OUT/prof_ingest_pmc1 $OUT/prof_ingest_pmc2
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -f csv -d $OUT/prof_ingest_pmc1 -o ingest -- $I > $OUT/prof_ingest_pmc1.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_ingest_pmc2 -o ingest -- $I > $OUT/prof_ingest_pmc2.log 2>&1) || true

# BASELINE config 2 (1M x 384, 256 queries, two lanes): the kernels of its step and the sweep's HBM traffic
C2="python $REPO/bench.py --only-config2 --config2-lane-sweep 2 --config2-batches 200 --oracle-queries 0"
rm -rf $OUT/prof_c2 $OUT/prof_c2_pmc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_c2 -o c2 -- $C2 > $OUT/prof_c2.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_c2_pmc -o c2 -- $C2 > $OUT/prof_c2_pmc.log 2>&1) || true
# the product-quantised engine's leg (N4): kernels of its step, and the scan's LDS counters
PQ="python $REPO/bench.py --only-pq --oracle-queries 0"
rm -rf $OUT/prof_pq $OUT/prof_pq_pmc
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_pq -o pq -- $PQ > $OUT/prof_pq.log 2>&1) || true
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/prof_pq_pmc -o pq -- $PQ > $OUT/prof_pq_pmc.log 2>&1) || true
# the non-uniform legs (clustered / anisotropic / gaussian): kernels of their steps
D="python $REPO/bench.py --only-distribution --oracle-queries 0"
rm -rf $OUT/prof_dist
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_dist -o dist -- $D > $OUT/prof_dist.log 2>&1) || true
# the traces themselves are large: only the summaries travel back
find $OUT/prof_* -name '*kernel_trace.csv' -size +8M -delete 2>/dev/null || true
