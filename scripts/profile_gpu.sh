#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then PMC passes in
# their OWN runs (never mixed with sys/runtime tracing).  Outputs land in gpurun_out/prof_*;
# scripts/summarize_profiles.py condenses them into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
REPO=$PWD
OUT=$REPO/gpurun_out
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-ingest"
rm -rf $OUT/prof_trace $OUT/prof_pmc1 $OUT/prof_pmc2 $OUT/prof_ingest
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_trace -o scan -- python $REPO/bench.py $ARGS > $OUT/prof_trace.log 2>&1) || true
(cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_pmc1 -o scan -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ingest > $OUT/prof_pmc1.log 2>&1) || true
(cd /tmp && timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -f csv -d $OUT/prof_pmc2 -o scan -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ingest > $OUT/prof_pmc2.log 2>&1) || true
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_ingest -o ingest -- python $REPO/scripts/ingest_bench.py --gib 8 > $OUT/prof_ingest.log 2>&1) || true
ls -R $OUT | head -50
