#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05p; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -15 | tee $o/tests.txt
for i in 1 2; do
for cfg in "--engine fp8_wgrad=0" "--engine fp8_wgrad=1"; do
  python bench.py --no-cpu-baseline --steps 6 --warmup 3 --experiment simclr --precision fp8 --per-gpu-batch 2048 $cfg 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$cfg]'.ljust(26), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done
python bench.py --no-cpu-baseline --steps 6 --warmup 3 --experiment simclr --per-gpu-batch 2048 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[bf16]'.ljust(26), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done | tee $o/ab_fp8.txt
