#!/bin/bash
# round 5: the matrix pipes' power ceiling for the e4m3 scaled MFMA next to bf16 (scripts/probes/mfma_power.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 scripts/probes/mfma_power > gpurun_out/r05_mfma_power_fp8.txt 2>&1
cat gpurun_out/r05_mfma_power_fp8.txt
