#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05o; mkdir -p $o
for i in 1 2 3; do
for v in noprio new pwg pr128; do
  lib=scripts/abl/lib$v.so; [ $v = new ] && lib=simhand_amd/libsimhand_hip.so
  SIMHAND_LIB=$lib python bench.py --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(10), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')}, d['device_state']['sclk_mhz']['mean'] if d['device_state'].get('available') else None)"
done; done | tee $o/ab_prio.txt
