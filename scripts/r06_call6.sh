#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py::test_rn50_handclr_w_bf16_at_the_reference_128px_geometry -m gpu -x -q --timeout=900 2>&1 | tail -3
python scripts/layer_table.py --alternates --out gpurun_out/cfg/r06_layer_table_alternates.md > /dev/null 2> gpurun_out/cfg/lt.err
grep -B1 "alt g1_k512" gpurun_out/cfg/r06_layer_table_alternates.md | cut -c1-200
