#!/bin/bash
cd "$(dirname "$0")/.."
o=gpurun_out/r05i; mkdir -p $o
for v in 0 8 11 12 13; do echo -n "lib ring$v (0 real, 8 per-tap all zero-page, 11 ring-A zero-page, 12 W zero-page, 13 both): "; SIMHAND_LIB=scripts/abl/libring$v.so timeout 200 python scripts/ring_abl.py 2>&1 | tail -1; done | tee $o/ring_abl.txt
