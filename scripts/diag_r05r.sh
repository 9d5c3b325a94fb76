#!/bin/bash
# round 5: the "thread" staging mode again after giving its helper thread a process group of its own + the per-mode test
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "staging_mode" 2>&1 | tail -5
python scripts/dist_stress.py --tag r05_thread_canary --staging thread --minutes 5 --groups 3 --diag --canary > /dev/null 2>&1
tail -n 1 gpurun_out/dist_stress_r05_thread_canary.log
grep -c " ok " gpurun_out/dist_stress_r05_thread_canary.log
