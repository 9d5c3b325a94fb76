#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05k; mkdir -p $o
for v in new valu3 valu6 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v (extra VALU per MFMA group: 0 / 3 / 6 -> +24 / +48 per k-step)" ; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
done 2>&1 | grep -v amdgpu.ids | tee $o/valu.txt
