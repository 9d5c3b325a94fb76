#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05h; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_backbone_ops.py -q -m gpu -k "big_tile or tile" 2>&1 | tail -6 | tee $o/tests.txt
for v in r04 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v" ; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
  for sh in "256 256 3 1 14" "512 512 3 1 7" "1024 256 1 1 14" "512 2048 1 1 7" "2048 512 1 1 7" "1024 512 1 1 14"; do
    SIMHAND_LIB=$lib timeout 120 python scripts/one_conv.py $sh 2048 20 2>&1 | tail -1
  done
done 2>&1 | grep -v amdgpu.ids | tee $o/ab.txt
