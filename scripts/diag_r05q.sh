#!/bin/bash
# round 5: energy decomposition of the tile loop's skeleton (scripts/probes/energy_parts.hip) + the fp8 tests added after the closing run's snapshot
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 scripts/probes/energy_parts > gpurun_out/r05_energy_parts.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu > gpurun_out/r05_fp8_tests.log 2>&1
tail -3 gpurun_out/r05_fp8_tests.log
cat gpurun_out/r05_energy_parts.txt
