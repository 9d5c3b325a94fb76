#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
o=gpurun_out/r05b; mkdir -p $o
rocprofv3 -L > $o/counters_all.txt 2>&1
grep -o "TCC_[A-Za-z0-9_]*\|TCP_[A-Za-z0-9_]*" $o/counters_all.txt | sort -u | tr '\n' ' ' > $o/counters_tc.txt
for v in base abl1 abl9 kord1; do
  lib=scripts/abl/lib$v.so; [ $v = base ] && lib=simhand_amd/libsimhand_hip.so
  echo "== $v" ; SIMHAND_LIB=$lib timeout 300 python scripts/tile_overhead.py 2>&1 | tail -5
done > $o/ablate.txt 2>&1
cat $o/ablate.txt
i=0
for v in base kord1; do
  lib=scripts/abl/lib$v.so; [ $v = base ] && lib=simhand_amd/libsimhand_hip.so
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
    i=$((i+1)); rm -rf /tmp/pm_$i
    echo "## $v : $set"
    SIMHAND_LIB=$lib timeout 240 rocprofv3 --pmc $set -d /tmp/pm_$i -o p -- python scripts/one_conv.py 256 256 3 1 14 2048 3 > /tmp/pm_$i.log 2>&1 || tail -3 /tmp/pm_$i.log
    python scripts/pmc_dump.py /tmp/pm_$i/p_results.db igemm256 2>&1 | tail -8
  done
done > $o/pmc_tc.txt 2>&1
cat $o/pmc_tc.txt
