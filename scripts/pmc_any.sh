#!/bin/bash
# On the GPU box: PMC counters of the kernels matching <pattern> while running an arbitrary python script.
# usage: bash scripts/pmc_any.sh <kernel-name-substring> <script.py> [args...]   (counter sets collected in separate passes)
pat=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set -d /tmp/pmc_$i -o p -- python "$@" > /tmp/pmc_$i.log 2>&1 || tail -5 /tmp/pmc_$i.log
  python scripts/pmc_dump.py /tmp/pmc_$i/p_results.db "$pat"
done
