#!/bin/bash
# round 5: whole-line output stores in igemm256_kernel's plain forward epilogue (abl53) vs the committed epilogue
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_fullline_stores.txt
: > $out
SIMHAND_LIB=scripts/abl/libabl53.so timeout 600 python -m pytest tests/test_gpu_backbone_ops.py -x -q -m gpu -k "big_tile or igemm256 or tile256" 2>&1 | tail -3 >> $out
for v in base abl53 base abl53; do
  echo "== $v" >> $out
  if [ $v = base ]; then L=""; else L="scripts/abl/lib$v.so"; fi
  SIMHAND_LIB=$L timeout 300 python scripts/tile_overhead.py 2>&1 | grep -v amdgpu.ids >> $out
  for shp in "256 256 3 1 14" "512 512 3 1 7" "1024 256 1 1 14" "512 2048 1 1 7"; do
    SIMHAND_LIB=$L timeout 120 python scripts/one_conv.py $shp 2048 20 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
