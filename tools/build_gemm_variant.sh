#!/bin/bash
# libaum_hip_<name>.so with gemm.hip alone rebuilt under extra flags (the other objects are the default build's: csrc/build.py first);
# seconds instead of the full build of tools/build_variant.sh.   tools/build_gemm_variant.sh <name> <flags ...>
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=audio-mamba-aum_amd/csrc
mkdir -p audio-mamba-aum_amd/aum_hip/variants /tmp/gv_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable "$@" \
    -c $C/gemm.hip -o /tmp/gv_$name/gemm.o
objs=$(ls $C/_obj/*.o | grep -v '/gemm.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$name.so $objs /tmp/gv_$name/gemm.o
echo built $name
