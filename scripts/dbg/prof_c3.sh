# Kernel trace of BASELINE config 3 (10M x 768 L2, 1024 queries: L2 on the int8 tier) + the HBM bytes of its filter launch.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; ulimit -c 0
REPO=$PWD; OUT=$REPO/gpurun_out
B="python $REPO/scripts/config_sweep.py"
rm -rf $OUT/prof_c3 $OUT/prof_c3_pmc
(cd /tmp && ONLY=C3 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/prof_c3 -o c3 -- $B > $OUT/prof_c3.log 2>&1) || true
(cd /tmp && ONLY=C3 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/prof_c3_pmc -o c3 -- $B > $OUT/prof_c3_pmc.log 2>&1) || true
tail -1 $OUT/prof_c3.log | cut -c1-700
f=$(find $OUT/prof_c3 -name '*kernel_stats.csv' | head -1); head -16 "$f" | cut -c1-200
