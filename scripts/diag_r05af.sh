#!/bin/bash
# round 5: chained 1x1 forward (conv3 + BN + residual + ReLU + next conv1) with its residual rows requested one chunk ahead vs the previous build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r05_res_ahead.txt
: > $out
timeout 900 python -m pytest tests/test_gpu_backbone_ops.py tests/test_gpu_configs.py -x -q -m gpu -k "chain or bnact or step or resnet" 2>&1 | tail -2 >> $out
for i in 1 2 3; do
for v in prev new; do
  if [ $v = new ]; then L=""; else L="scripts/abl/libprev.so"; fi
  SIMHAND_LIB=$L python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(10), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')}, d['device_state']['sclk_mhz']['mean'])" >> $out
done; done
cat $out
