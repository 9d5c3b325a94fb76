#!/bin/bash
# round 5: BatchNorm + residual + ReLU epilogue of the 256 x 256 kernel with its residual rows requested unconditionally vs the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_ep256.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_backbone_ops.py -x -q -m gpu -k "bnact or ep or epilogue or residual" 2>&1 | tail -2 >> $out
for v in prev new prev new; do
  echo "== $v" >> $out
  if [ $v = new ]; then L=""; else L="scripts/abl/libprev.so"; fi
  SIMHAND_LIB=$L timeout 300 python scripts/ep256_bench.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out
