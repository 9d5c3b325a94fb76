#!/bin/bash
# Variant builds of the 256x256 kernel's DMA issue schedule (conv_igemm.hip, SH_DMA_SCHED=n) into scripts/abl/libdma_N.so; run HERE.
set -e
cd "$(dirname "$0")/../simhand_amd/csrc"
mkdir -p ../../scripts/abl build
for n in ${SET:-1 2 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSH_DMA_SCHED=$n ${EXTRA:-} -c conv_igemm.hip -o build/v_igemm_$n.o
  objs=$(ls build/*.o | grep -v conv_igemm | grep -v v_igemm | grep -v abl)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/v_igemm_$n.o -ldl -o ../../scripts/abl/libdma_$n.so
done
ls -la ../../scripts/abl
