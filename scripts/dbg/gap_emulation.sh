#!/bin/bash
# VERDICT r4 #6, the kill criterion: what would the two-lane step cost if (a) the sample pass + tau selection + collect
# kernel were no launches of their own (folded into the head of the persistent sweep) and (b) a tighter int8 bound let the
# fp64 re-score look at ~280 instead of 384 candidates per query?  The measurement build skips / shortens them
# (YAMS_ACCEL_EMU_NO_SAMPLE, YAMS_ACCEL_EMU_KPRIME in scan_api.cpp) and keeps the sweep; one query batch repeated so that
# the stale thresholds are the right ones.  Prints ms_per_step of every form; the fold itself would ADD the sample tiles'
# share of the sweep (1/64 of the rows: ~0.11 ms) plus a grid barrier to the emulated number.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/r5c}; mkdir -p $OUT
export YAMS_ACCEL_MEASURE_LIB=1
B="python bench.py --steps 20 --warmup 5 --query-batches 1 --no-cpu-baseline --no-ingest --no-hbm-leg --no-l2-leg --no-c-abi-leg --no-boundary-leg --no-telemetry --oracle-queries 0"
run() { # name, env...
  local name=$1; shift
  for rep in 1 2; do
    env "$@" $B --extra-json $OUT/emu_${name}_$rep.json 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', $rep, 'ms_per_step', j['ms_per_step'], 'launch_ms', j['roofline']['launch_ms'])"
  done
}
run baseline X=1
run no_sample YAMS_ACCEL_EMU_NO_SAMPLE=1
run kprime280 YAMS_ACCEL_EMU_KPRIME=280
run no_sample_kprime280 YAMS_ACCEL_EMU_NO_SAMPLE=1 YAMS_ACCEL_EMU_KPRIME=280
run no_sample_kprime200 YAMS_ACCEL_EMU_NO_SAMPLE=1 YAMS_ACCEL_EMU_KPRIME=200
