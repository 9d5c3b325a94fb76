#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
python -m pytest tests/test_gpu_backbone_ops.py -k "n64 or n128" tests/test_gpu_merged_launches.py::test_rn50_step_is_bit_identical_with_fewer_launches -m gpu -x -q --timeout=900 2>&1 | tail -4
python scripts/l1_fold_dgrad_bench.py 2>&1 | tail -12
bash scripts/ab_generic.sh "" "--switch N128=5" 3 > gpurun_out/cfg/r06_n64_ab.txt 2>&1
cat gpurun_out/cfg/r06_n64_ab.txt
