#!/bin/bash
# Diagnostic build of libmpcodec.so with in-kernel phase stamps (mp_pursuit.hip: MPC_STAMPS); load it with
#   MPCODEC_LIB=imageexperiments_amd/lib/libmpcodec_stamps.so python tools/quick_bench.py raise 1
# Every persistent launch then prints the average cycles a wave spends per phase and step to stderr (s_memtime: shader clocks --
# against the workgroup residency span, which is s_memrealtime at 100 MHz, they give the clock the kernel actually ran at:
# 1.78 GHz on 256 workgroups, 2.37 GHz on 64, profiles/r03_stamps.txt).  The product build contains no stamp.
set -e
cd "$(dirname "$0")/../imageexperiments_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -DMPC_STAMPS \
    -shared -o ../lib/libmpcodec_stamps.so host_dictionary.cpp host_bitstream.cpp host_codec.cpp host_stats.cpp mp_kernels.hip mp_pursuit.hip mp_streams.hip mp_entropy.hip mpcodec_capi.cpp mpcodec_multi.cpp
echo ../lib/libmpcodec_stamps.so
