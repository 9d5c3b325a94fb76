#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05m; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_backbone_ops.py -q -m gpu -k "big_tile" 2>&1 | tail -3
for v in stag1 stag2; do
  echo "== correctness with $v"; SIMHAND_LIB=scripts/abl/lib$v.so timeout 600 python -m pytest tests/test_gpu_backbone_ops.py -q -m gpu -k "big_tile" 2>&1 | tail -3
done
for v in new prio3 stag1 stag2 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v"; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
  for sh in "256 256 3 1 14" "512 512 3 1 7" "1024 256 1 1 14" "512 2048 1 1 7"; do
    SIMHAND_LIB=$lib timeout 120 python scripts/one_conv.py $sh 2048 20 2>&1 | tail -1
  done
done 2>&1 | grep -v amdgpu.ids | tee $o/stagger.txt
