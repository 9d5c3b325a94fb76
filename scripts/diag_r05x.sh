#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
SIMHAND_LIB=scripts/abl/libabl31.so timeout 300 python scripts/stamp256_tile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_tile_stamps.txt
cat gpurun_out/r05_tile_stamps.txt
