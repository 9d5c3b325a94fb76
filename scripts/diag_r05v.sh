#!/bin/bash
# round 5: first-round start skew in igemm256_kernel (SH_SKEW256 = s_sleep units per CU index step) vs none
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_skew256.txt
: > $out
for v in base skew3 skew6 skew12 base skew6; do
  echo "== $v" >> $out
  if [ $v = base ]; then L=""; else L="scripts/abl/lib$v.so"; fi
  SIMHAND_LIB=$L timeout 300 python scripts/tile_overhead.py 2>&1 | grep -v amdgpu.ids >> $out
  for shp in "256 256 3 1 14" "512 512 3 1 7" "1024 256 1 1 14" "512 2048 1 1 7"; do
    SIMHAND_LIB=$L timeout 120 python scripts/one_conv.py $shp 2048 20 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
