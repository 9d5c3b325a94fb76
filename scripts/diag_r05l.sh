#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05l; mkdir -p $o
for v in new prio1 prio2 prio3 prio4 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v"; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
done 2>&1 | grep -v amdgpu.ids | tee $o/prio.txt
for n in 1 2 3 4; do echo "== stamps prio$n"; SIMHAND_LIB=scripts/abl/libsprio$n.so timeout 200 python scripts/stamp256.py 2>&1 | grep -v amdgpu.ids | head -11; done | tee $o/prio_stamps.txt
