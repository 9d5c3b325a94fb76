#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/cfg
{
for v in plain masked merge step; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/g1t; rocprofv3 --pmc $c -d /tmp/g1t -o f -- python scripts/g1_dgrad_traffic.py $v > /tmp/g1t.log 2>&1 || tail -3 /tmp/g1t.log
    echo "## $v $c"; python scripts/pmc_dump.py /tmp/g1t/f_results.db gemm1x1 | grep -v "^==" 
  done
done
} > gpurun_out/cfg/r06_g1_dgrad_traffic.txt 2>&1
cat gpurun_out/cfg/r06_g1_dgrad_traffic.txt
