#!/bin/bash
# Variant build of ONE source file into scripts/abl/lib<name>.so (A/B timing on one GPU box: bench.py --lib PATH / simhand_amd._lib.set_library_paths); run HERE.
# usage: scripts/build_variant.sh <name> <file.hip> <extra hipcc flags...>
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../simhand_amd/csrc"
mkdir -p ../../scripts/abl build
base=${src%.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src -o build/var_${name}.o
objs=$(ls build/*.o | grep -v "/${base}.o" | grep -v "/var_" | grep -v "/v_" | grep -v abl)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var_${name}.o -ldl -o ../../scripts/abl/lib${name}.so
ls -la ../../scripts/abl/lib${name}.so
