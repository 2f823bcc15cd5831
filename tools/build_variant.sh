#!/bin/bash
# a variant of libmpcodec.so for same-box A/B timing: tools/build_variant.sh <name> [extra hipcc flags ...]  -> ab_libs/libmpcodec_<name>.so
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ab_libs
cd $R/imageexperiments_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread "$@" -shared -o $R/ab_libs/libmpcodec_$name.so \
    host_dictionary.cpp host_bitstream.cpp host_codec.cpp host_stats.cpp mp_kernels.hip mp_pursuit.hip mp_streams.hip mp_entropy.hip mpcodec_capi.cpp mpcodec_multi.cpp
echo ab_libs/libmpcodec_$name.so
