#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the profile set of ONE bench.py configuration other than the headline one -- kernel-trace stats, the two
# HBM-traffic PMC passes (own passes, program directly after --), the per-layer table at that geometry, and the bench line with its roofline.
# usage: bash scripts/profile_config.sh <tag> <images for the layer table> <layer-table args> -- <bench.py args>
#   e.g. bash scripts/profile_config.sh r06_rn152_b512 1024 "--resnet 152" -- --resnet 152 --experiment peclr_w --per-gpu-batch 512
# Output: gpurun_out/cfg/<tag>_{kernel_stats,hbm_traffic,layer_table}.md, <tag>.json (copy the ones to be judged into profiles/).
set -u
tag=$1; images=$2; lt_args=$3; shift 3
[ "$1" = "--" ] && shift
root=$(pwd)
out=$root/gpurun_out/cfg
mkdir -p "$out"
export TMPDIR=/tmp
rm -rf /tmp/pc_kt /tmp/pc_f /tmp/pc_w
rocprofv3 --kernel-trace --stats -d /tmp/pc_kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-probe "$@" > /tmp/pc_kt.log 2>&1
python scripts/rocpd_stats.py /tmp/pc_kt/kt_results.db "$out/${tag}_kernel_stats.md" > /dev/null || tail -5 /tmp/pc_kt.log
rocprofv3 --pmc FETCH_SIZE -d /tmp/pc_f -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-probe "$@" > /tmp/pc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/pc_w -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-probe "$@" > /tmp/pc_w.log 2>&1
python scripts/pmc_traffic.py /tmp/pc_f/f_results.db /tmp/pc_w/w_results.db "$out/${tag}_hbm_traffic.md" "$out/${tag}_hbm_traffic.json" > /dev/null || tail -5 /tmp/pc_f.log
python scripts/layer_table.py --images "$images" $lt_args --out "$out/${tag}_layer_table.md" > /dev/null 2> /tmp/pc_lt.log || tail -5 /tmp/pc_lt.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" 2> /tmp/pc_b.log | tail -1 > "$out/${tag}.json" || tail -5 /tmp/pc_b.log
python - "$out/${tag}.json" <<'EOF'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"{sys.argv[1]}: {d['ms_per_step']:.2f} ms/step, {d['value']:.0f} pairs/s, dominant class {r['kernel']} {r['achieved']:.0f} TFLOP/s = {r['frac']:.3f} of {r['peak']:.0f}; "
      f"whole step {r['step_tflops_per_gpu']:.0f} TFLOP/s")
EOF
ls -la "$out" | tail -8
