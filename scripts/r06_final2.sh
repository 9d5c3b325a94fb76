#!/bin/bash
# round 6: the profile passes once more without the parity probe's launches in the tables (bench.py --no-parity-probe), then the bench line
set -u
mkdir -p gpurun_out/cfg gpurun_out/refresh
export TMPDIR=/tmp
REFRESH_FAST=1 bash scripts/refresh_profiles.sh r06 > gpurun_out/refresh_r06b.log 2>&1
tail -4 gpurun_out/refresh_r06b.log | cut -c1-300
bash scripts/profile_config.sh r06_rn50_128px 6272 "--image-size 128" -- --image-size 128 --per-gpu-batch 3136
bash scripts/profile_config.sh r06_fp8_b2048 4096 "" -- --experiment simclr --precision fp8 --per-gpu-batch 2048
bash scripts/profile_config.sh r06_rn152_b512 1024 "--resnet 152" -- --resnet 152 --experiment peclr_w --per-gpu-batch 512
