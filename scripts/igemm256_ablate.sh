#!/bin/bash
# Ablation builds of the tile kernel's 256x256 kernel's DMA stream (conv_igemm.hip, SH_ABL256=n): which resource binds the MFMA-bound layers?
# Run HERE to build scripts/abl/libabl_N.so, then on the GPU box: python scripts/igemm_ablate.py
set -e
cd "$(dirname "$0")/../simhand_amd/csrc"
mkdir -p ../../scripts/abl build
for n in ${ABL_SET:-0 4 5}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSH_ABL256=$n -c conv_igemm.hip -o build/conv_igemm_abl$n.o
  objs=$(ls build/*.o | grep -v conv_igemm)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/conv_igemm_abl$n.o -ldl -o ../../scripts/abl/libabl_$n.so
done
ls -la ../../scripts/abl
