#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
export TMPDIR=/tmp
./scripts/probes/mfma_power > gpurun_out/cfg/r06_mfma_power.md 2>&1
cat gpurun_out/cfg/r06_mfma_power.md
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline > /tmp/tl.log 2>&1
python scripts/timeline.py /tmp/tl/tl_results.db gpurun_out/cfg/r06_step_timeline_before.txt
head -1 gpurun_out/cfg/r06_step_timeline_before.txt
