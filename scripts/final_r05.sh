#!/bin/bash
# round-5 closing run on the GPU box: shared-GPU stress in the three non-default staging modes, profile refresh, full GPU suite
cd "$(dirname "$0")/.."
python scripts/dist_stress.py --tag r05_thread_canary --staging thread --minutes 4 --groups 3 --diag --canary > /dev/null 2>&1
python scripts/dist_stress.py --tag r05_buckets --staging buckets --minutes 3 --groups 3 --diag > /dev/null 2>&1
python scripts/dist_stress.py --tag r05_off_canary --staging off --minutes 4 --groups 3 --diag --canary > /dev/null 2>&1
tail -1 gpurun_out/dist_stress_r05_*.log
bash scripts/refresh_profiles.sh r05 > gpurun_out/refresh_r05.log 2>&1
tail -3 gpurun_out/refresh_r05.log | cut -c1-400
python -m pytest tests -q -m gpu 2>&1 | tail -6 > gpurun_out/r05_gputests_final.log; cat gpurun_out/r05_gputests_final.log
