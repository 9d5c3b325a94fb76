#!/bin/bash
# round 5: e4m3 data gradient of the 256 x 256 kernel without its in-loop register spill (the second-segment code compiled out) vs the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_fp8_dgrad_nospill.txt
: > $out
timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -2 >> $out
for v in prev new prev new; do
  echo "== $v" >> $out
  if [ $v = new ]; then L=""; else L="scripts/abl/libprev.so"; fi
  SIMHAND_LIB=$L timeout 300 python scripts/fp8_conv_bench.py 4096 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
