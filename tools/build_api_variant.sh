#!/bin/bash
# libaum_hip_<name>.so with the api object alone (AUM_API_PART=3: conv / norm / frontend / sums kernels + the extern "C" surface) rebuilt under
# extra flags; the other objects are the default build's (csrc/build.py first).   tools/build_api_variant.sh <name> <flags ...>
set -e
name=$1; shift
cd "$(dirname "$0")/.."
C=audio-mamba-aum_amd/csrc
mkdir -p audio-mamba-aum_amd/aum_hip/variants /tmp/av_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable -DAUM_API_PART=3 "$@" \
    -c $C/aum_hip.hip -o /tmp/av_$name/api.o
objs=$(ls $C/_obj/*.o | grep -v '/api.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$name.so $objs /tmp/av_$name/api.o
echo built $name
