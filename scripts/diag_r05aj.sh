#!/bin/bash
# round 5: the shared-GPU irregularity with every instrument on -- staging off, canaries, collective audit, and the step run twice per rank (LOCALDIFF)
cd "$GRAFT_REPO_ROOT" || exit 1
python scripts/dist_stress.py --tag r05_off_rerun --staging off --minutes 25 --groups 3 --diag --canary --rerun > /dev/null 2>&1
tail -n 1 gpurun_out/dist_stress_r05_off_rerun.log
grep -c " ok " gpurun_out/dist_stress_r05_off_rerun.log
grep -n "LOCALDIFF\|RERUN.*[1-9][0-9]* collective\|FAILED" gpurun_out/dist_stress_r05_off_rerun.log | head -40 | cut -c1-600
