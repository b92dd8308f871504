#!/bin/bash
# build libaum_hip.so with extra compile flags (e.g. -DAUM_ABLATE for the kbench ablation bits, which the production library does not contain) into audio-mamba-aum_amd/aum_hip/variants/libaum_hip_<name>.so (A/B runs; AUM_DEBUG=1 AUM_HIP_LIB=...)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p audio-mamba-aum_amd/aum_hip/variants
AUM_EXTRA_CXXFLAGS="$*" python audio-mamba-aum_amd/csrc/build.py --force > /tmp/build_$name.log 2>&1 || { tail -20 /tmp/build_$name.log; exit 1; }
cp audio-mamba-aum_amd/aum_hip/libaum_hip.so audio-mamba-aum_amd/aum_hip/variants/libaum_hip_$name.so
echo built $name
