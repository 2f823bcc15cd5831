// mp_streams.hip -- product: device-side stream assembly for the MI355X tile encoder.
//
// The container holds, besides `lengths`, 6K symbol streams: codes[2K*ch + 2i] = deltaId and [+1] = intCoeff of step i of
// channel ch for every tile with more than i atoms, tiles in the reference's visiting order (x outer, y inner:
// CompressedImage.cpp:535-572); the three step-0 coefficient streams are difference coded (:428-446).  The pursuit leaves
// records [tile][3][K] of which most are dead (97 MB for a 16 Mpixel K = 32 frame, 21 MB alive).  These kernels compact the
// live symbols stream by stream, in stream order, into ONE contiguous u16 buffer laid out in container order and apply the
// DC differencing, so that only the live symbols cross PCIe and every host coding job starts from a finished stream:
//   count    per block of 1024 tiles and channel: tiles with more than i atoms, for every i (from a histogram of the counts)
//   scan     exclusive scan of those over the blocks (one wave per stream pair); stream sizes
//   scatter  stream offsets from the sizes; rank of every live symbol inside its block by ballots, write deltaId / intCoeff to
//            their stream positions
//   dc       zig-zag difference of the three step-0 coefficient streams (CompressedImage.cpp:428-446)
// HBM-bound byte shuffling: reads are 16-byte vectors of whole records, writes are runs of consecutive u16.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mp_device.h"

namespace mpc {

namespace {
constexpr int kBlockTiles = 1024;               // tiles per workgroup (thread = tile)
constexpr int kWavesPerBlock = kBlockTiles / 64;
}  // namespace

__global__ __launch_bounds__(kBlockTiles) void mp_stream_count_kernel(const StreamArgs a)
{
    __shared__ unsigned hist[3][kMaxDeviceK + 2];
    for (int i = threadIdx.x; i < 3 * (kMaxDeviceK + 2); i += kBlockTiles) (&hist[0][0])[i] = 0;
    __syncthreads();
    const long long t = (long long)blockIdx.x * kBlockTiles + threadIdx.x;
    if (t < a.tiles) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int c = a.counts[t * 3 + ch];
            atomicAdd(&hist[ch][c < a.K ? c : a.K], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 * a.K) {
        const int ch = threadIdx.x / a.K, i = threadIdx.x - ch * a.K;
        unsigned live = 0;
        for (int c = i + 1; c <= a.K; ++c) live += hist[ch][c];
        a.block_live[((long long)blockIdx.x * 3 + ch) * a.K + i] = live;
    }
}

// one wave per (channel, step): exclusive scan of the live counts over the blocks, in place; the stream's size
__global__ __launch_bounds__(64) void mp_stream_scan_kernel(const StreamArgs a, int blocks)
{
    const int pair = blockIdx.x, ch = pair / a.K, i = pair - ch * a.K, lane = threadIdx.x;
    unsigned run = 0;
    for (int base = 0; base < blocks; base += 64) {
        const int b = base + lane;
        const long long at = ((long long)(b < blocks ? b : 0) * 3 + ch) * a.K + i;
        const unsigned n = b < blocks ? a.block_live[at] : 0u;
        unsigned incl = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (b < blocks) a.block_live[at] = run + incl - n;            // live count -> offset of the block in its stream
        run += __shfl(incl, 63);
    }
    if (lane == 0) a.sizes[pair] = run;
}

// the streams in container order: codes[2K*ch + 2i] = deltaId, [+1] = intCoeff, each sizes[ch*K + i] symbols
__device__ __forceinline__ void stream_layout(const StreamArgs& a, unsigned long long* off /*[6K + 1], LDS*/)
{
    if (threadIdx.x == 0) {
        unsigned long long at = 0;
        for (int p = 0; p < 3 * a.K; ++p) {
            const unsigned n = a.sizes[p];
            off[2 * p] = at;
            at += n;
            off[2 * p + 1] = at;
            at += n;
        }
        off[6 * a.K] = at;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlockTiles) void mp_stream_scatter_kernel(const StreamArgs a)
{
    __shared__ unsigned wave_live[kWavesPerBlock][kMaxDeviceK];   // live tiles of the waves in front of this one, per step
    __shared__ unsigned block_at[3 * kMaxDeviceK];                // the block's first position in each stream
    __shared__ unsigned long long off[6 * kMaxDeviceK + 1];
    stream_layout(a, off);
    if (blockIdx.x == 0)                                          // for the kernels and copies that follow
        for (int s = threadIdx.x; s <= 6 * a.K; s += kBlockTiles) a.stream_off[s] = off[s];
    if ((int)threadIdx.x < 3 * a.K) block_at[threadIdx.x] = a.block_live[(long long)blockIdx.x * 3 * a.K + threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t = (long long)blockIdx.x * kBlockTiles + threadIdx.x;
    const bool in = t < a.tiles;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    for (int ch = 0; ch < 3; ++ch) {
        const int c = in ? min((int)a.counts[t * 3 + ch], a.K) : 0;
        // only the live part of the record is read: a lane asks for the 16-byte pieces that hold its first c steps (most
        // tile-channels stop after a few steps, and the dead steps are three quarters of the records' bytes)
        uint32_t rec[kMaxDeviceK];
        {
            const uint4* src = reinterpret_cast<const uint4*>(a.choices + ((in ? t : 0) * 3 + ch) * a.K);
#pragma unroll
            for (int v = 0; v < kMaxDeviceK / 4; ++v) {
                uint4 x = make_uint4(0, 0, 0, 0);
                if (4 * v < c && (a.K & 3) == 0) x = src[v];
                rec[4 * v] = x.x; rec[4 * v + 1] = x.y; rec[4 * v + 2] = x.z; rec[4 * v + 3] = x.w;
            }
            if ((a.K & 3) != 0) {                                // K not a multiple of 4: records are not 16-byte aligned
                const uint32_t* s1 = a.choices + ((in ? t : 0) * 3 + ch) * a.K;
#pragma unroll
                for (int i = 0; i < kMaxDeviceK; ++i) rec[i] = i < c ? s1[i] : 0u;
            }
        }
        __syncthreads();                                          // wave_live of the previous channel is no longer read
#pragma unroll
        for (int i = 0; i < kMaxDeviceK; ++i)
            if (i < a.K) {
                const unsigned long long live = __ballot(c > i);
                if (lane == 0) wave_live[wave][i] = (unsigned)__popcll(live);
            }
        __syncthreads();
        if ((int)threadIdx.x < a.K) {                             // counts -> exclusive prefix over the waves, once per step
            unsigned run = 0;
#pragma unroll
            for (int w = 0; w < kWavesPerBlock; ++w) {
                const unsigned n = wave_live[w][threadIdx.x];
                wave_live[w][threadIdx.x] = run;
                run += n;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kMaxDeviceK; ++i)
            if (i < a.K) {
                const unsigned long long live = __ballot(c > i);
                if (!live) continue;
                if (c > i) {
                    const unsigned long long pos = (unsigned long long)block_at[ch * a.K + i] + wave_live[wave][i] + (unsigned)__popcll(live & below);
                    const unsigned long long od = off[2 * (ch * a.K + i)], oc = off[2 * (ch * a.K + i) + 1];
                    a.symbols[od + pos] = (uint16_t)(rec[i] & 0xFFFFu);
                    // step-0 coefficients go through the difference kernel: parked behind the end of all streams
                    uint16_t* coeff = i == 0 ? a.dc_tmp + (long long)ch * a.tiles : a.symbols + oc;
                    coeff[pos] = (uint16_t)(rec[i] >> 16);
                }
            }
    }
}

// The inverse of the scatter, for the decoder (CompressedImage.cpp:680-705 read the streams back into per-tile records): the same
// positions -- block offset + rank inside the block -- but the symbols are READ from the 6K streams (`symbols`, laid out as the
// scatter writes them; the step-0 coefficients already summed up by the host) and every tile's records written whole, dead steps
// as zeros.
__global__ __launch_bounds__(kBlockTiles) void mp_stream_gather_kernel(const StreamArgs a, uint32_t* __restrict__ choices)
{
    __shared__ unsigned wave_live[kWavesPerBlock][kMaxDeviceK];
    __shared__ unsigned long long off[6 * kMaxDeviceK + 1];
    stream_layout(a, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long t = (long long)blockIdx.x * kBlockTiles + threadIdx.x;
    const bool in = t < a.tiles;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    for (int ch = 0; ch < 3; ++ch) {
        const int c = in ? (int)a.counts[t * 3 + ch] : 0;
        uint32_t rec[kMaxDeviceK];
        __syncthreads();                                          // wave_live of the previous channel is no longer read
#pragma unroll
        for (int i = 0; i < kMaxDeviceK; ++i)
            if (i < a.K) {
                const unsigned long long live = __ballot(c > i);
                if (lane == 0) wave_live[wave][i] = (unsigned)__popcll(live);
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kMaxDeviceK; ++i) {
            rec[i] = 0u;
            if (i < a.K) {
                const unsigned long long live = __ballot(c > i);
                if (!live) continue;
                unsigned before = 0;
                for (int w = 0; w < wave; ++w) before += wave_live[w][i];
                if (c > i) {
                    const unsigned long long pos = (unsigned long long)a.block_live[((long long)blockIdx.x * 3 + ch) * a.K + i] + before +
                                                   (unsigned)__popcll(live & below);
                    const unsigned long long od = off[2 * (ch * a.K + i)], oc = off[2 * (ch * a.K + i) + 1];
                    rec[i] = (uint32_t)a.symbols[od + pos] | ((uint32_t)a.symbols[oc + pos] << 16);
                }
            }
        }
        if (in) {
            uint32_t* dst = choices + (t * 3 + ch) * a.K;
            if ((a.K & 3) == 0) {
#pragma unroll
                for (int v = 0; v < kMaxDeviceK / 4; ++v)
                    if (4 * v < a.K) reinterpret_cast<uint4*>(dst)[v] = make_uint4(rec[4 * v], rec[4 * v + 1], rec[4 * v + 2], rec[4 * v + 3]);
            } else {
#pragma unroll
                for (int i = 0; i < kMaxDeviceK; ++i)
                    if (i < a.K) dst[i] = rec[i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void mp_stream_dc_kernel(const StreamArgs a)
{
    for (int ch = 0; ch < 3; ++ch) {
        const unsigned n = a.sizes[ch * a.K];
        const uint16_t* src = a.dc_tmp + (long long)ch * a.tiles;
        uint16_t* dst = a.symbols + a.stream_off[2 * (ch * a.K) + 1];
        for (unsigned long long j = (unsigned long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (unsigned long long)gridDim.x * 256) {
            const int32_t v = src[j], prev = j ? (int32_t)src[j - 1] : 0;
            const int32_t d = v - prev;
            dst[j] = (uint16_t)(((uint32_t)d << 1) ^ (uint32_t)(d >> 31));     // zigzagEncode, BitBuffer.h:116
        }
    }
}

// A row stripe's records (tile t = tx * rows + ty_local, the order mpc_encode_tiles_device writes a stripe in) copied to their
// places in a whole frame's records (t = tx * tiles_y + row_begin + ty_local): the step between the stripe exchange of the
// multi-GPU path and the stream assembly, which reads whole frames in the reference's tile order (CompressedImage.cpp:535-537).
// Both sides are runs of rows * 3K words per tile column: coalesced reads and writes.
__global__ __launch_bounds__(256) void mp_interleave_stripe_kernel(const uint16_t* __restrict__ part_counts, const uint32_t* __restrict__ part_choices,
                                                                   int tiles_x, int tiles_y, int row_begin, int rows, int K,
                                                                   uint16_t* __restrict__ frame_counts, uint32_t* __restrict__ frame_choices)
{
    const long long words_per_col = (long long)rows * 3 * K, n_words = words_per_col * tiles_x;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) {
        const long long tx = i / words_per_col, w = i - tx * words_per_col;
        frame_choices[(tx * tiles_y + row_begin) * 3 * K + w] = part_choices[i];
    }
    const long long halves_per_col = (long long)rows * 3, n_halves = halves_per_col * tiles_x;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_halves; i += (long long)gridDim.x * 256) {
        const long long tx = i / halves_per_col, w = i - tx * halves_per_col;
        frame_counts[(tx * tiles_y + row_begin) * 3 + w] = part_counts[i];
    }
}

int launch_interleave_stripe(const uint16_t* part_counts, const uint32_t* part_choices, int tiles_x, int tiles_y, int row_begin, int rows,
                             int K, uint16_t* frame_counts, uint32_t* frame_choices, void* stream_)
{
    if (tiles_x < 1 || rows < 1 || row_begin < 0 || row_begin + rows > tiles_y || K < 1 || K > kMaxDeviceK) return (int)hipErrorInvalidValue;
    const long long n_words = (long long)rows * 3 * K * tiles_x;
    const unsigned blocks = (unsigned)((n_words + 255) / 256 < 4096 ? (n_words + 255) / 256 : 4096);
    hipLaunchKernelGGL(mp_interleave_stripe_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream_), part_counts, part_choices,
                       tiles_x, tiles_y, row_begin, rows, K, frame_counts, frame_choices);
    return (int)hipGetLastError();
}

int launch_stream_gather(const StreamArgs& a, uint32_t* choices, void* stream_)
{
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int blocks = (int)((a.tiles + kBlockTiles - 1) / kBlockTiles);
    if (blocks < 1 || a.K < 1 || a.K > kMaxDeviceK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mp_stream_count_kernel, dim3((unsigned)blocks), dim3(kBlockTiles), 0, s, a);
    hipLaunchKernelGGL(mp_stream_scan_kernel, dim3((unsigned)(3 * a.K)), dim3(64), 0, s, a, blocks);
    hipLaunchKernelGGL(mp_stream_gather_kernel, dim3((unsigned)blocks), dim3(kBlockTiles), 0, s, a, choices);
    return (int)hipGetLastError();
}

size_t stream_workspace_words(long long tiles, int K)
{
    const long long blocks = (tiles + kBlockTiles - 1) / kBlockTiles;
    return (size_t)(blocks * 3 * K);
}

int launch_stream_assembly(const StreamArgs& a, void* stream_)
{
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int blocks = (int)((a.tiles + kBlockTiles - 1) / kBlockTiles);
    if (blocks < 1 || a.K < 1 || a.K > kMaxDeviceK) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(mp_stream_count_kernel, dim3((unsigned)blocks), dim3(kBlockTiles), 0, s, a);
    hipLaunchKernelGGL(mp_stream_scan_kernel, dim3((unsigned)(3 * a.K)), dim3(64), 0, s, a, blocks);
    hipLaunchKernelGGL(mp_stream_scatter_kernel, dim3((unsigned)blocks), dim3(kBlockTiles), 0, s, a);
    hipLaunchKernelGGL(mp_stream_dc_kernel, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace mpc
