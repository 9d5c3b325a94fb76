#!/bin/bash
# round 6 closing run ON THE GPU BOX: both test suites, the headline profile set, the profile sets of the other configurations
set -u
mkdir -p gpurun_out/cfg gpurun_out/refresh
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --timeout=900 -x > gpurun_out/r06_gputests_final.log 2>&1
tail -3 gpurun_out/r06_gputests_final.log
bash scripts/refresh_profiles.sh r06 > gpurun_out/refresh_r06.log 2>&1
tail -25 gpurun_out/refresh_r06.log | cut -c1-400
bash scripts/profile_config.sh r06_rn50_128px 6272 "--image-size 128" -- --image-size 128 --per-gpu-batch 3136
bash scripts/profile_config.sh r06_fp8_b2048 4096 "" -- --experiment simclr --precision fp8 --per-gpu-batch 2048
bash scripts/profile_config.sh r06_rn152_b512 1024 "--resnet 152" -- --resnet 152 --experiment peclr_w --per-gpu-batch 512
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-parity-probe > /tmp/tl.log 2>&1
python scripts/timeline.py /tmp/tl/tl_results.db gpurun_out/cfg/r06_step_timeline.txt
head -1 gpurun_out/cfg/r06_step_timeline.txt
