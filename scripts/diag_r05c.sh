#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05c; mkdir -p $o
for v in r04 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v" ; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
done > $o/slope.txt 2>&1
cat $o/slope.txt
timeout 900 python -m pytest tests/test_gpu_backbone_ops.py tests/test_gpu_fullsize.py tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -8 | tee $o/tests.txt
for v in r04 new; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v"; SIMHAND_LIB=$lib timeout 600 python scripts/layer_table.py --out $o/lt_$v.md > /dev/null 2>&1; grep igemm256 $o/lt_$v.md
done > $o/lt.txt 2>&1
cat $o/lt.txt
