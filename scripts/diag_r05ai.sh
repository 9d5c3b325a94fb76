#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/dist_stress.py --tag r05_rerun_smoke --staging off --minutes 2 --groups 1 --max-reps 3 --diag --canary --rerun > /dev/null 2>&1
cat gpurun_out/dist_stress_r05_rerun_smoke.log | cut -c1-400 | head -60
