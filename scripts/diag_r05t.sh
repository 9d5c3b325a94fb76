#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python scripts/tile_overhead_fp8.py > gpurun_out/r05_tile_overhead_fp8.txt 2>&1
cat gpurun_out/r05_tile_overhead_fp8.txt
