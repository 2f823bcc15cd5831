#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void touch(unsigned* p, size_t n) { size_t i = blockIdx.x * 256ull + threadIdx.x; if (i < n) p[i] += 1; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const size_t bytes = 6600000 / 4 * 4, n = bytes / 4;
    unsigned *d, *h;
    CK(hipMalloc(&d, bytes)); CK(hipHostMalloc(&h, bytes, mode == 5 ? hipHostMallocMapped : hipHostMallocDefault));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int it = 0; it < 10; ++it) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, a, d, n);
        if (mode == 0) {                        // same stream as the kernel
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a));
            CK(hipStreamSynchronize(a));
        } else if (mode == 1) {                 // a stream that never runs kernels, behind an event
            CK(hipEventRecord(ev, a)); CK(hipStreamWaitEvent(b, ev, 0));
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b));
            CK(hipStreamSynchronize(b));
        } else if (mode == 2) {                 // batch API with the overlap flag
            void* dsts[1] = {h}; void* srcs[1] = {d}; size_t sizes[1] = {bytes}; size_t idx[1] = {0}; size_t fail = 0;
            hipMemcpyAttributes at{}; at.srcAccessOrder = hipMemcpySrcAccessOrderStream; at.flags = hipMemcpyFlagPreferOverlapWithCompute;
            at.srcLocHint.type = hipMemLocationTypeDevice; at.dstLocHint.type = hipMemLocationTypeHost;
            CK(hipMemcpyBatchAsync(dsts, srcs, sizes, 1, &at, idx, 1, &fail, a));
            CK(hipStreamSynchronize(a));
        } else if (mode == 5) {                 // mapped pinned memory, same stream
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, a));
            CK(hipStreamSynchronize(a));
        } else if (mode == 6) {                 // the null stream
            hipLaunchKernelGGL(touch, dim3((n + 255) / 256), dim3(256), 0, nullptr, d, n);
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, nullptr));
            CK(hipStreamSynchronize(nullptr));
        } else if (mode == 3) {                 // host waits, then a synchronous copy
            CK(hipStreamSynchronize(a));
            CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        } else if (mode == 4) {                 // host waits, then async on the copy-only stream
            CK(hipStreamSynchronize(a));
            CK(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, b));
            CK(hipStreamSynchronize(b));
        }
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (it >= 8) std::printf("mode %d: %.3f ms, h[5]=%u\n", mode, ms, h[5]);
    }
    return 0;
}
