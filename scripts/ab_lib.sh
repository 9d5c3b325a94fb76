#!/bin/bash
# same-box A/B of variant builds of the library (scripts/build_variant.sh -> scripts/abl/lib<name>.so): alternating bench.py runs, per-class kernel time.
# usage (GPU box): bash scripts/ab_lib.sh [rounds] <name> [<name> ...]     ("" = the in-tree library is always the first of each round)
# e.g.  scripts/build_variant.sh g1nt conv_1x1.hip -DSOME_MACRO=1 ; gpurun -- 'bash scripts/ab_lib.sh 3 g1nt'
R=${1:-3}; shift
for i in $(seq 1 "$R"); do
  for n in "" "$@"; do
    v=""; [ -n "$n" ] && v="--lib scripts/abl/lib${n}.so"
    python bench.py --no-cpu-baseline --no-parity-probe --steps 8 --warmup 3 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[${n:-in-tree}]'.ljust(24), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
  done
done
