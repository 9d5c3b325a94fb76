#!/bin/bash
# round 5: LDS buffering / occupancy variants of the generic tile kernel on the layer-1 folded data gradient, and in the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_igemm128_variants.txt
: > $out
for v in base nbuf2 minb2 minb4 base; do
  echo "== $v" >> $out
  if [ $v = base ]; then L=""; else L="scripts/abl/lib$v.so"; fi
  SIMHAND_LIB=$L timeout 300 python scripts/l1_fold_dgrad_bench.py 2>&1 | grep -v amdgpu.ids >> $out
  SIMHAND_LIB=$L python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(10), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')}, d['device_state']['sclk_mhz']['mean'])" >> $out
done
cat $out
