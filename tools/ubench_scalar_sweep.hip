// ubench_scalar_sweep.hip -- diagnostic microbenchmark (not part of the product library).
// Measures shader clocks per atom of the base sweep's inner loop (s_load-fed v_mul_f64/v_add_f64 chain)
// for different grid sizes and dictionary footprints, to separate VALU issue time from scalar-cache latency.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_scalar_sweep.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef const double __attribute__((address_space(4))) * cptr_t;

template <int SLOTS>
__global__ __launch_bounds__(64) void sweep(const double* __restrict__ base_g, int natoms, int wrap, double* outv,
                                            unsigned long long* cycles)
{
    double r[SLOTS][64];
#pragma unroll
    for (int s = 0; s < SLOTS; s++)
#pragma unroll
        for (int j = 0; j < 64; j++) r[s][j] = (double)((threadIdx.x * 7 + j * 3 + s) % 31) - 15.0;
    cptr_t base = (cptr_t)(uintptr_t)base_g;
    double best[SLOTS];
    for (int s = 0; s < SLOTS; s++) best[s] = 0.0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int a = 0; a < natoms; a++) {
        cptr_t row = base + (size_t)(a % wrap) * 64;
        double tot[SLOTS];
#pragma unroll
        for (int s = 0; s < SLOTS; s++) tot[s] = 0.0;
#pragma unroll
        for (int j = 0; j < 64; j++)
#pragma unroll
            for (int s = 0; s < SLOTS; s++) tot[s] += row[j] * r[s][j];
#pragma unroll
        for (int s = 0; s < SLOTS; s++)
            if (__builtin_fabs(tot[s]) > __builtin_fabs(best[s])) best[s] = tot[s];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double acc = 0;
    for (int s = 0; s < SLOTS; s++) acc += best[s];
    outv[blockIdx.x * 64 + threadIdx.x] = acc;
    if (threadIdx.x == 0) atomicAdd(cycles, t1 - t0);
}

// hand-pipelined variant: two 16-double scalar groups; the next group's s_load is issued right after the
// wait for the current group (after its first MAC), so exactly one group load is in flight while 15 MACs run.
__device__ __forceinline__ void load16(double (&g)[16], cptr_t p) {
#pragma unroll
    for (int i = 0; i < 16; i++) g[i] = p[i];
}
#define MAC_GROUP(cur, nxt, jbase, nextptr)                      \
    tot += cur[0] * r[jbase];                                     \
    __builtin_amdgcn_sched_barrier(0);                            \
    load16(nxt, nextptr);                                         \
    __builtin_amdgcn_sched_barrier(0);                            \
    _Pragma("unroll") for (int i = 1; i < 16; i++) tot += cur[i] * r[jbase + i]; \
    __builtin_amdgcn_sched_barrier(0);

__global__ __launch_bounds__(64) void sweep_pipe(const double* __restrict__ base_g, int natoms, int wrap, double* outv,
                                                 unsigned long long* cycles, int desync)
{
    const int mode = desync >> 4;     // 0 normal, 1 no argmax (sum only), 2 fused multiply-add (NOT the reference arithmetic)
    desync &= 15;
    double r[64];
#pragma unroll
    for (int j = 0; j < 64; j++) r[j] = (double)((threadIdx.x * 7 + j * 3) % 31) - 15.0;
    cptr_t base = (cptr_t)(uintptr_t)base_g;
    double best = 0.0;
    unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st0 = __builtin_amdgcn_s_memtime();
    double ga[16], gb[16];
    load16(ga, base);
    for (int a = 0; a < natoms; a++) {
        const int shift = desync ? (int)(blockIdx.x * 37u) : 0;      // desync: every wave walks the rows from its own offset
        cptr_t row = base + (size_t)((a + shift) % wrap) * 64;
        cptr_t nxt = base + (size_t)((a + 1 + shift) % wrap) * 64;
        double tot = 0.0;
        MAC_GROUP(ga, gb, 0, row + 16)
        MAC_GROUP(gb, ga, 16, row + 32)
        MAC_GROUP(ga, gb, 32, row + 48)
        MAC_GROUP(gb, ga, 48, nxt)
        if (mode == 1) best += tot;
        else if (__builtin_fabs(tot) > __builtin_fabs(best)) best = tot;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long st1 = __builtin_amdgcn_s_memtime();
    outv[blockIdx.x * 64 + threadIdx.x] = best;
    if (threadIdx.x == 0) atomicAdd(cycles, t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 7) { cycles[1] = st1 - st0; cycles[2] = rt1 - rt0; }   // shader clocks vs 100 MHz ticks
}

void run_pipe(const double* d_base, int natoms, int wrap, int waves, double* d_out, unsigned long long* d_cyc, int desync = 0)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(sweep_pipe, dim3(waves), dim3(64), 0, 0, d_base, natoms, wrap, d_out, d_cyc, desync);
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(sweep_pipe, dim3(waves), dim3(64), 0, 0, d_base, natoms, wrap, d_out, d_cyc, desync);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc[3]; hipMemcpy(cyc, d_cyc, 24, hipMemcpyDeviceToHost);
    printf("PIPE%s waves=%5d wrap=%4d atoms=%d : %.3f ms, %.0f clk/atom/wave (ideal 512), MAC-lanes/s = %.2f T, shader clock %.0f MHz\n", desync ? "(desync)" : "        ", waves, wrap,
           natoms, ms, (double)cyc[0] / waves / natoms, (double)waves * 64 * 64.0 * natoms / (ms * 1e-3) / 1e12,
           cyc[2] ? 100.0 * (double)cyc[1] / (double)cyc[2] : 0.0);
}

template <int SLOTS>
void run(const double* d_base, int natoms, int wrap, int waves, double* d_out, unsigned long long* d_cyc)
{
    hipMemset(d_cyc, 0, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(sweep<SLOTS>, dim3(waves), dim3(64), 0, 0, d_base, natoms, wrap, d_out, d_cyc);  // warm
    hipMemset(d_cyc, 0, 8);
    hipEventRecord(e0);
    hipLaunchKernelGGL(sweep<SLOTS>, dim3(waves), dim3(64), 0, 0, d_base, natoms, wrap, d_out, d_cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
    double per_atom = (double)cyc / waves / natoms;
    printf("slots=%d waves=%5d wrap=%4d atoms=%d : %.3f ms, %.0f clk/atom/wave (ideal %d), MAC-lanes/s = %.2f T\n", SLOTS, waves,
           wrap, natoms, ms, per_atom, 512 * SLOTS, (double)waves * 64 * SLOTS * 64.0 * natoms / (ms * 1e-3) / 1e12);
}

int main()
{
    const int rows = 512;
    std::vector<double> h(rows * 64);
    for (size_t i = 0; i < h.size(); i++) h[i] = ((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    double* d_base; double* d_out; unsigned long long* d_cyc;
    hipMalloc(&d_base, h.size() * 8); hipMemcpy(d_base, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&d_out, 8192 * 64 * 8); hipMalloc(&d_cyc, 64);
    const int natoms = 5100;
    for (int wrap : {510, 16}) {
        for (int waves : {2048, 3072}) run_pipe(d_base, natoms, wrap, waves, d_out, d_cyc);
        for (int waves : {3072}) run_pipe(d_base, natoms, wrap, waves, d_out, d_cyc, 16);

    }
    return 0;
}
