#!/bin/bash
# round-5 diagnostic A: what the 256-wide tile kernel's k-loop waits for (ablation builds, garbage results) + LDS counters
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05a
for v in base abl4 abl6 abl7 abl8; do
  lib=scripts/abl/lib$v.so; [ $v = base ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v" ; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
done > gpurun_out/r05a/ablate.txt 2>&1
cat gpurun_out/r05a/ablate.txt
timeout 600 bash scripts/pmc_any.sh igemm256 scripts/tile_overhead.py > gpurun_out/r05a/pmc_base.txt 2>&1
cat gpurun_out/r05a/pmc_base.txt
