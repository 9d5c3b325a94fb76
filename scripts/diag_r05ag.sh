#!/bin/bash
# round 5: a longer attempt to reproduce the shared-GPU irregularity in the arrangement that showed it (staging off), canaries + audit on
cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/dist_stress.py --tag r05_off_canary_long --staging off --minutes 16 --groups 3 --diag --canary > /dev/null 2>&1
tail -n 1 gpurun_out/dist_stress_r05_off_canary_long.log
grep -c " ok " gpurun_out/dist_stress_r05_off_canary_long.log
