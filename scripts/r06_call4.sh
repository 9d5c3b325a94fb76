#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
export TMPDIR=/tmp
python -m pytest tests/test_gpu_merged_launches.py tests/test_gpu_dist.py -m gpu -x -q --timeout=900 > gpurun_out/gputests_b1.log 2>&1
tail -5 gpurun_out/gputests_b1.log
python -m pytest tests -m gpu -x -q --timeout=900 --durations=12 --deselect tests/test_gpu_merged_launches.py --deselect tests/test_gpu_dist.py > gpurun_out/gputests_b2.log 2>&1
tail -20 gpurun_out/gputests_b2.log
rm -rf /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline > /tmp/tl.log 2>&1
python scripts/timeline.py /tmp/tl/tl_results.db gpurun_out/cfg/r06_step_timeline_merged.txt
head -1 gpurun_out/cfg/r06_step_timeline_merged.txt
bash scripts/ab_generic.sh "" "--switch FOLD_LEGACY=1" 3 > gpurun_out/cfg/r06_merged_launches_ab.txt 2>&1
cat gpurun_out/cfg/r06_merged_launches_ab.txt
