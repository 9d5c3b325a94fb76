#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/cfg
python -m pytest tests/test_gpu_backbone_ops.py -k "1x1 or dgrad or shortcut" tests/test_gpu_fullsize.py -m gpu -x -q --timeout=900 2>&1 | tail -3
{
for v in merge step; do
  for c in FETCH_SIZE; do
    rm -rf /tmp/g1t; rocprofv3 --pmc $c -d /tmp/g1t -o f -- python scripts/g1_dgrad_traffic.py $v > /tmp/g1t.log 2>&1 || tail -3 /tmp/g1t.log
    echo "## (staged masks) $v $c"; python scripts/pmc_dump.py /tmp/g1t/f_results.db gemm1x1 | grep -v "^=="
  done
done
} >> gpurun_out/cfg/r06_g1_dgrad_traffic.txt 2>&1
tail -4 gpurun_out/cfg/r06_g1_dgrad_traffic.txt
{
for i in 1 2 3; do
for v in "--lib scripts/abl/libg1old.so" ""; do
python bench.py --no-cpu-baseline --no-parity-probe --steps 8 --warmup 3 $v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_step']; print('[$v]'.ljust(40), round(d['ms_per_step'],2), {a: round(b,2) for a,b in k.items() if a in ('conv_fwd','conv_dgrad','conv_wgrad','bn','misc')})"
done; done
} > gpurun_out/cfg/r06_g1_staged_masks_ab.txt 2>&1
cat gpurun_out/cfg/r06_g1_staged_masks_ab.txt
