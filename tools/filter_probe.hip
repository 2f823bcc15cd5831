// filter_probe.hip -- diagnostic harness (not part of the product library): runs mp_filter_kernel /
// mp_detail_filter_kernel alone on a fabricated workspace and prints time per launch and, with the phase stamps
// compiled in, where thread 0 of each block spends its clocks.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMPC_STAMPS -Iimageexperiments_amd/csrc \
//         tools/filter_probe.hip imageexperiments_amd/csrc/host_dictionary.cpp -o tools/filter_probe
//   tools/filter_probe [tile-channels per channel = 16200] [blocks = 768] [with_detail0 = 1]
#include "../imageexperiments_amd/csrc/mp_kernels.hip"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "host_dictionary.h"

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

template <class T>
static T* to_device(const std::vector<T>& v)
{
    T* d = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d), v.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}

int main(int argc, char** argv)
{
    const int per_channel = argc > 1 ? std::atoi(argv[1]) : 16200;
    const int blocks = argc > 2 ? std::atoi(argv[2]) : 768;
    const int with_detail0 = argc > 3 ? std::atoi(argv[3]) : 1;
    const int n = 3 * per_channel;
    const int K = 8;

    const mpc::Dictionary dict = mpc::build_dictionary(8);
    mpc::DictDevice dd{};
    int padded = 0;
    dd.base = to_device(mpc::base_padded(dict, 2, &padded));
    dd.num_base = dict.num_base;
    dd.base_rows_padded = padded;
    const size_t det_rows = static_cast<size_t>(dict.total_detail_rows());
    std::vector<double> det((3 * det_rows + 1) * 64, 0.0);
    for (int ch = 0; ch < 3; ++ch) std::copy(dict.detail[ch].begin(), dict.detail[ch].end(), det.begin() + ch * det_rows * 64);
    dd.detail = to_device(det);
    dd.detail_rows = static_cast<long long>(det_rows);
    dd.block_rows = to_device(dict.block_rows);
    dd.block0_rows = dict.block_rows[0];
    dd.block_row_off = to_device(dict.block_row_off);
    dd.base_f32 = to_device(mpc::filter_tiles(dict.base.data(), dict.num_base, mpc::kBaseFilterTiles));
    std::vector<uint16_t> det32;
    for (int ch = 0; ch < 3; ++ch)
        for (int b = 0; b < dict.num_base; ++b) {
            const std::vector<uint16_t> t = mpc::filter_tiles(dict.detail[ch].data() + static_cast<size_t>(dict.block_row_off[b]) * 64,
                                                           dict.block_rows[b], mpc::kBlockFilterTiles);
            det32.insert(det32.end(), t.begin(), t.end());
        }
    dd.detail_f32 = to_device(det32);

    void* mem = nullptr;
    CHECK(hipMalloc(&mem, mpc::workspace_bytes(n, K)));
    CHECK(hipMemset(mem, 0, mpc::workspace_bytes(n, K)));
    const mpc::Workspace ws = mpc::carve_workspace(mem, n, K);

    // residuals of a "second step": zero-mean noise (a few survivors per tile-channel, like real residuals)
    std::mt19937 rng(1);
    std::normal_distribution<double> noise(0.0, 20.0);
    std::vector<double> r(static_cast<size_t>(n) * 64);
    for (double& v : r) v = noise(rng);
    CHECK(hipMemcpy(ws.r, r.data(), r.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> ids(per_channel), nblk(n, 0x101);
    for (int ch = 0; ch < 3; ++ch) {
        for (int i = 0; i < per_channel; ++i) ids[i] = i * 3 + ch;
        CHECK(hipMemcpy(ws.act[0][ch], ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    CHECK(hipMemcpy(ws.nblk, nblk.data(), nblk.size() * sizeof(int), hipMemcpyHostToDevice));
    const unsigned counters[16] = {(unsigned)per_channel, (unsigned)per_channel, (unsigned)per_channel};
    CHECK(hipMemcpy(ws.counters, counters, sizeof(counters), hipMemcpyHostToDevice));

    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 20;
    for (int pass = 0; pass < 2; ++pass) {
#ifdef MPC_STAMPS
        unsigned long long zero[16] = {0};
        CHECK(hipMemcpyToSymbol(HIP_SYMBOL(mpc::g_stamps), zero, sizeof(zero)));
#endif
        CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < reps; ++i)
            hipLaunchKernelGGL(mpc::mp_filter_kernel, dim3(blocks), dim3(256), 0, nullptr, mpc::filter_args(ws, dd, 0), 0, with_detail0);
        CHECK(hipEventRecord(e1, nullptr));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass == 0) continue;
        const int units = 3 * ((per_channel + 15) / 16);
        std::printf("filter kernel: %.1f us per launch, %d units of 16 tile-channels, %d blocks -> %.2f us per unit per block slot\n",
                    1000.0 * ms / reps, units, blocks, 1000.0 * ms / reps / ((units + blocks - 1) / blocks));
#ifdef MPC_STAMPS
        unsigned long long st[16];
        CHECK(hipMemcpyFromSymbol(st, HIP_SYMBOL(mpc::g_stamps), sizeof(st)));
        static const char* names[11] = {"stage (ds_write of prefetched rows)", "barrier", "MFMA phase (+prefetch issue)", "barrier",
                                        "scan", "barrier", "survivors (exact)", "combine", "barrier", "merge + store", "barrier"};
        double total = 0.0;
        for (int k = 0; k < 11; ++k) total += (double)st[k];
        for (int k = 0; k < 11; ++k)
            std::printf("  %-38s %8.0f clocks per unit  %5.1f %%\n", names[k], (double)st[k] / reps / units, 100.0 * st[k] / total);
        std::printf("  total %.0f clocks per unit\n", total / reps / units);
#endif
    }
    return 0;
}
