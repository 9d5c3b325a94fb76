#!/bin/bash
# round 5: what the per-tile intercept of igemm256_kernel consists of (epilogue ablations: no stores / streaming stores / no BN sums / neither)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_intercept_ablation.txt
: > $out
for v in base abl5 abl50 abl51 abl52 base abl50; do
  echo "== $v" >> $out
  if [ $v = base ]; then timeout 300 python scripts/tile_overhead.py >> $out 2>&1
  else SIMHAND_LIB=scripts/abl/lib$v.so timeout 300 python scripts/tile_overhead.py >> $out 2>&1; fi
done
cat $out
