#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests/test_gpu_backbone_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fp8.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
