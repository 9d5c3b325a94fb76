#!/bin/bash
set -u
mkdir -p gpurun_out/cfg
{
for i in 1 2 3; do
for v in "" "--lib scripts/abl/libg1minb2.so"; do
python bench.py --no-cpu-baseline --no-parity-probe --steps 8 --warmup 3 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(40), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done; done
} > gpurun_out/cfg/r06_g1_minb2_ab.txt 2>&1
cat gpurun_out/cfg/r06_g1_minb2_ab.txt
