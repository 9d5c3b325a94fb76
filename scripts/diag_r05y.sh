#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 120 scripts/probes/store_pattern > gpurun_out/r05_store_pattern.txt 2>&1
cat gpurun_out/r05_store_pattern.txt
