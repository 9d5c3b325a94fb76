#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05n; mkdir -p $o
for v in new prio2 prio3 prio6 prio8 prio9 prio10 new prio3; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v"; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
  for sh in "256 256 3 1 14" "1024 256 1 1 14" "512 2048 1 1 7"; do
    SIMHAND_LIB=$lib timeout 120 python scripts/one_conv.py $sh 2048 20 2>&1 | tail -1
  done
done 2>&1 | grep -v amdgpu.ids | tee $o/prio2.txt
