#!/bin/bash
# On the GPU box: memory-path PMC counters (SQ / TA / TCP / TCC, separate passes, each under its own timeout: some counter
# combinations stall rocprofv3 on this pool) of the kernels matching <pattern>.
# usage: bash scripts/pmc_mem.sh <kernel-name-substring> <script.py> [args...]
pat=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  timeout 150 rocprofv3 --pmc $set -d /tmp/pmcm_$i -o p -- python "$@" > /tmp/pmcm_$i.log 2>&1 || { echo "pass $i failed / timed out: $set"; tail -2 /tmp/pmcm_$i.log; continue; }
  python scripts/pmc_dump.py /tmp/pmcm_$i/p_results.db "$pat"
done
