#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for w in 1 2 3 1 2 3; do
  echo "window $w: $(YAMS_ACCEL_I8R_WINDOW=$w timeout 200 python scripts/filter_ablation.py i8:2 2>/dev/null | tail -1)"
done
for w in 1 2; do echo "window $w $(YAMS_ACCEL_I8R_WINDOW=$w PMC_ONLY=5 bash scripts/pmc_groups.sh i8:2 2>&1 | tail -1 | grep -o "FETCH_SIZE.*")"; done
