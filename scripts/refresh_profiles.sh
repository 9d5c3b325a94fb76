#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): bench line + rocprofv3 kernel-trace stats + two PMC passes -> gpurun_out/refresh/.
# usage: gpurun --timeout 1500 -- 'bash scripts/refresh_profiles.sh r01'
#        REFRESH_FAST=1 ...: only what ties the bench line to the sources -- kernel-trace stats, the two HBM-traffic passes, the bench line
#        (about 3 of the 7 minutes; the matrix-core pass, the layer table and the other configurations keep their previous files)
# Afterwards (here): cp gpurun_out/refresh/* profiles/
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/refresh
rm -rf "$out" && mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity-probe > /tmp/kt.log 2>&1
python scripts/rocpd_stats.py /tmp/prof_kt/kt_results.db "$out/${tag}_bench_b1024_kernel_stats.md" > /dev/null
rocprofv3 --pmc FETCH_SIZE -d /tmp/prof_f -o f -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-probe > /tmp/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d /tmp/prof_w -o w -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-probe > /tmp/w.log 2>&1
python scripts/pmc_traffic.py /tmp/prof_f/f_results.db /tmp/prof_w/w_results.db "$out/${tag}_bench_b1024_hbm_traffic.md" "$out/hbm_traffic.json" > /dev/null
fast=${REFRESH_FAST:-0}
# matrix-core utilisation (own pass: counters only, program directly after --)
[ "$fast" = 1 ] || rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/prof_m -o m -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-parity-probe > /tmp/m.log 2>&1
[ "$fast" = 1 ] || python scripts/pmc_mfma.py /tmp/prof_m/m_results.db "$out/${tag}_bench_b1024_mfma_util.md" /tmp/prof_kt/kt_results.db > /dev/null || tail -5 /tmp/m.log
[ "$fast" = 1 ] || python scripts/layer_table.py --out "$out/${tag}_layer_table.md" > /dev/null 2>&1
tail -n 3 /tmp/f.log | cut -c1-200; tail -n 3 /tmp/w.log | cut -c1-200
# the bench line last: its roofline.traffic reads the PMC-derived bytes per launch written just above
cp "$out/hbm_traffic.json" "$root/profiles/hbm_traffic.json"
python bench.py --cpu-full > "$out/${tag}_bench_b1024.log" 2>&1
tail -1 "$out/${tag}_bench_b1024.log" > "$out/${tag}_bench_b1024.json"
cat "$out/${tag}_bench_b1024.json"
[ "$fast" = 1 ] && { ls -la "$out"; exit 0; }
# the other configurations of the same step on this box (one line each): simclr bf16 vs fp8 (BASELINE configs[4] arithmetic; also at its
# per-GPU batch of 2048 pairs = 4096 images), precision 16
# (fp16 storage + GradScaler: the reference's policy), ResNet-18 / ResNet-152, and the reference's own 128 x 128 geometry (training_config.json:38-41)
# at a matched pixel count (3136 pairs) and at the headline's pair count
{
  for cfg in "--experiment simclr" "--experiment simclr --precision fp8" "--experiment simclr --per-gpu-batch 2048" "--experiment simclr --precision fp8 --per-gpu-batch 2048" "--precision 16" "--resnet 18" "--resnet 152 --experiment peclr_w --per-gpu-batch 512" "--image-size 128 --per-gpu-batch 3136" "--image-size 128"; do
    python bench.py --steps 8 --warmup 3 --no-cpu-baseline $cfg 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('| bench.py $cfg |', round(d['ms_per_step'], 2), 'ms/step |', round(d['value']), 'pairs/s |', d['dtype'], '|', round(d['roofline']['step_tflops_per_gpu']), 'TFLOP/s whole step =', round(d['roofline']['step_tflops_per_gpu'] / d['roofline']['peak'], 3), 'of', round(d['roofline']['peak']), '| dominant class', d['roofline']['kernel'], round(d['roofline']['frac'], 3), '|')"
  done
} > "$out/${tag}_other_configs.md"
cat "$out/${tag}_other_configs.md"
ls -la "$out"
