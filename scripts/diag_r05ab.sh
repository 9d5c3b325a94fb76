#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/dgrad_epi_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_dgrad_epi_base.txt
cat gpurun_out/r05_dgrad_epi_base.txt
