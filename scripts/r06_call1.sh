#!/bin/bash
# round 6, GPU call 1: the MFMA-shape probe + profile sets of the other BASELINE configurations and the reference's 128x128 geometry
set -u
mkdir -p gpurun_out/cfg
./scripts/probes/mfma_power > gpurun_out/cfg/r06_mfma_power.md 2>&1
cat gpurun_out/cfg/r06_mfma_power.md
bash scripts/profile_config.sh r06_rn50_128px 6272 "--image-size 128" -- --image-size 128 --per-gpu-batch 3136
bash scripts/profile_config.sh r06_fp8_b2048 4096 "" -- --experiment simclr --precision fp8 --per-gpu-batch 2048
bash scripts/profile_config.sh r06_rn152_b512 1024 "--resnet 152" -- --resnet 152 --experiment peclr_w --per-gpu-batch 512
{
  for cfg in "--image-size 128" "--image-size 128 --per-gpu-batch 3136" "--experiment simclr --per-gpu-batch 2048" ""; do
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('| bench.py $cfg |', round(d['ms_per_step'], 2), 'ms/step |', round(d['value']), 'pairs/s |', d['dtype'], '|', round(d['roofline']['step_tflops_per_gpu']), 'TFLOP/s whole step |')"
  done
} > gpurun_out/cfg/r06_other_configs.md
cat gpurun_out/cfg/r06_other_configs.md
