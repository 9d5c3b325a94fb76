#!/bin/bash
# On the GPU box: PMC counters of one conv layer's kernel.  usage: bash scripts/pmc_conv.sh <kernel-name-substring> cin cout k stride h [N] [which]
# (counters collected in separate passes; no tracing options besides the implicit kernel dispatch records)
pat=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $set -d /tmp/pmc_$i -o p -- python scripts/one_conv.py "$@" > /tmp/pmc_$i.log 2>&1 || tail -5 /tmp/pmc_$i.log
  python scripts/pmc_dump.py /tmp/pmc_$i/p_results.db "$pat"
done
