// mp_kernels.hip -- product: hand-written gfx950 (CDNA4, wave64) kernels for the
// quantized matching-pursuit tile encoder.  No MFMA, no atomics on the data
// path, IEEE double with separately rounded mul/add (-ffp-contract=off): the
// integer outputs must equal the reference's double path exactly.
//
// Replaces, per 8x8 tile and channel (a "tile-channel"):
//   tile gather + YUVFromRGB      CompressedImage.cpp:538-554, misc.cpp:12-21
//   CalcMPDynamic                 MatchingPursuit.cpp:39-74
//   Select (GEMV + argmax)        MatchingPursuit.cpp:7-25, mathmatrix.cpp:426-449
//   dynamicBasis                  CompressedImage.cpp:212-250 (index table, no copy)
//   Scale / Subtract              mathvector.cpp:116-148
//
// Mapping (see DESIGN.md "Kernel"):
//   * one workgroup = one wave64; each LANE owns one tile-channel: its 64-double
//     residual lives in 128 VGPRs for the whole pursuit;
//   * base sweep ("mode A"): the 510 base atoms are shared by every tile-channel,
//     so the wave walks them once per step and every lane correlates the same
//     atom against its own residual.  Atom coefficients are wave-uniform: they
//     arrive through the scalar cache (s_load_dwordx16) and feed v_mul_f64 as an
//     SGPR operand -- no VGPRs, no LDS, no vector-memory traffic for the
//     dictionary.  The dot product is the reference's sequential tot += l*r.
//   * detail sweep ("mode B"): the KLT blocks unlocked by earlier choices differ
//     per tile-channel.  For those the wave turns to one tile-channel at a time:
//     its residual is staged in LDS (batches of 16 lanes), LANE = atom row of the
//     62/63-row block (transposed, coalesced 1 KiB loads), the residual element
//     is an LDS broadcast, and a 6-step shuffle butterfly takes the argmax with
//     lowest-index-wins ties (only the argmax is reduced across lanes -- never a
//     dot product, whose rounding order must stay sequential).
//   * lanes are persistent: a finished lane pulls the next tile-channel from a
//     global ticket counter (Y tile-channels first, they run longest), so lanes
//     at different MP steps of different tiles share every base sweep.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mp_device.h"

namespace mpc {

namespace {

typedef const double __attribute__((address_space(4))) * scalar_f64_ptr;   // constant address space: forces s_load
typedef const int32_t __attribute__((address_space(4))) * scalar_i32_ptr;

constexpr int N = 64;                  // pixels per tile
constexpr int STAGE_LANES = 16;        // tile-channels staged in LDS at a time for mode B

// YUV weights, ImageHelper/inc/misc.h:7-11; folded in double exactly like the reference's macros
constexpr double W_R = 0.299, W_G = 0.587, W_B = 0.114;
constexpr double U_SCALE = 0.436 / (1.0 - 0.114);
constexpr double V_SCALE = 0.615 / (1.0 - 0.299);

__device__ __forceinline__ void load_group(double (&g)[16], scalar_f64_ptr src) {
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = src[i];
}

// one 16-coefficient group of the sequential dot product: tot += atom[j] * r[j], j ascending
#define MAC_GROUP(cur, nxt, jbase, nextptr)                                              \
    tot += cur[0] * r[jbase];                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    load_group(nxt, nextptr);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    _Pragma("unroll") for (int i = 1; i < 16; ++i) tot += cur[i] * r[(jbase) + i];       \
    __builtin_amdgcn_sched_barrier(0);

__device__ __forceinline__ int wave_bcast_i32(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }

}  // namespace

__global__ __launch_bounds__(64, 2) void mp_encode_kernel(const EncodeParams p)
{
    __shared__ __attribute__((aligned(16))) double stage[N / 2][STAGE_LANES][2];   // 8 KiB: residuals, [pair][lane16][2]
    __shared__ uint16_t blocks[kMaxDeviceK][64];                                      // 4 KiB: per-lane chosen base atoms

    const int lane = threadIdx.x;
    const int K = p.K;
    const bool vector_mode = p.vec_in != nullptr;
    const int ntiles = p.tiles_x * p.tile_rows;
    const int nunits = ntiles * p.frames;                 // tiles of the whole batch
    const int n_y = vector_mode ? p.vec_count : nunits;   // tasks in the Y queue (vector mode: all of them)
    const int n_uv = vector_mode ? 0 : 2 * nunits;        // tasks in the chroma queue
    bool y_dry = false, uv_dry = (n_uv == 0);
    unsigned long long cyc[4] = {0, 0, 0, 0};
    unsigned iterations = 0;
    const scalar_f64_ptr base_s = (scalar_f64_ptr)(uintptr_t)p.base;
    const scalar_i32_ptr rows_s = (scalar_i32_ptr)(uintptr_t)p.block_rows;
    const scalar_i32_ptr rowoff_s = (scalar_i32_ptr)(uintptr_t)p.block_row_off;

    double r[N];
#pragma unroll
    for (int j = 0; j < N; ++j) r[j] = 0.0;

    int task = -1;          // ticket of the tile-channel this lane is working on, -1 = idle
    int ch = 0;             // channel of the task
    int out_index = 0;      // tile*3 + ch
    int step = 0;
    int prev_id = 0;
    int nblk = 0;           // entries in blocks[.][lane]
    int extra_rows = 0;     // rows appended after the base part (duplicates included)
    unsigned swept = 0;
    bool exhausted = false;

    for (;;) {
        // ------------------------------------------------------------------ refill idle lanes
        unsigned long long t_phase = 0;
        if (p.phase_cycles) t_phase = __builtin_readcyclecounter();
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const bool want = (task < 0) && !exhausted;
            if (!__ballot(want)) break;
            // queue choice: preferred one unless it is known dry
            const bool prefer_y = lane < p.y_lanes;
            const bool from_y = prefer_y ? !y_dry : uv_dry;
            const bool from_uv = !from_y;
            if (want && ((from_y && y_dry) || (from_uv && uv_dry))) exhausted = true;     // both queues dry
            const bool ask = want && !exhausted;
            const unsigned long long my = __ballot(ask && from_y);
            const unsigned long long muv = __ballot(ask && from_uv);
            int first_y = 0, first_uv = 0;
            if (lane == 0) {
                if (my) first_y = (int)atomicAdd(&p.queue[0], (unsigned)__popcll(my));
                if (muv) first_uv = (int)atomicAdd(&p.queue[1], (unsigned)__popcll(muv));
            }
            first_y = wave_bcast_i32(first_y, 0);
            first_uv = wave_bcast_i32(first_uv, 0);
            const unsigned long long below = (1ULL << lane) - 1ULL;
            int ticket = -1;
            if (ask) {
                if (from_y) {
                    const int t = first_y + (int)__popcll(my & below);
                    if (t < n_y) ticket = t;                       // Y queue: ticket = unit
                } else {
                    const int t = first_uv + (int)__popcll(muv & below);
                    if (t < n_uv) ticket = n_y + t;                // chroma queue: U and V of a tile are neighbours
                }
            }
            if (__ballot(ask && from_y && ticket < 0)) y_dry = true;      // wave-uniform knowledge
            if (__ballot(ask && from_uv && ticket < 0)) uv_dry = true;
            {
                {
                    if (ticket >= 0) {
                        task = ticket;
                        step = 0; prev_id = 0; nblk = 0; extra_rows = 0; swept = 0;
                        if (vector_mode) {
                            ch = p.vec_channel;
                            out_index = ticket;
                            const double* v = p.vec_in + (long long)ticket * N;
#pragma unroll
                            for (int j = 0; j < N; ++j) {
                                r[j] = v[j];
                                if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
                        int unit;
                        if (ticket < n_y) { ch = 0; unit = ticket; }
                        else { ch = 1 + ((ticket - n_y) & 1); unit = (ticket - n_y) >> 1; }
                        const int frame = unit / ntiles;
                        const int tile = unit - frame * ntiles;
                        const int tx = tile / p.tile_rows;
                        const int ty = p.tile_row_begin + (tile - tx * p.tile_rows);
                        out_index = unit * 3 + ch;
                        const uint8_t* frame_rgb = p.rgb + (long long)frame * p.frame_stride;
                        const int x0 = tx * 8, y0 = ty * 8;
#pragma unroll
                        for (int dy = 0; dy < 8; ++dy) {
                            const int v = y0 + dy;
                            const int vc = v < p.height ? v : p.height - 1;      // clamped address, value masked below
                            const uint8_t* line = frame_rgb + (long long)vc * p.row_stride;
#pragma unroll
                            for (int dx = 0; dx < 8; ++dx) {
                                const int u = x0 + dx;
                                const int uc = u < p.width ? u : p.width - 1;
                                const uint8_t* px = line + 3 * uc;
                                const double red = (double)px[0], green = (double)px[1], blue = (double)px[2];
                                const double Y = (W_R * red + W_G * green + W_B * blue);
                                // u = U_SCALE*(blue - Y), v = V_SCALE*(red - Y): operands selected, one mul (branch-free)
                                const double chroma = ((ch == 1) ? U_SCALE : V_SCALE) * (((ch == 1) ? blue : red) - Y);
                                const double val = (ch == 0) ? Y : chroma;
                                r[dx + 8 * dy] = (u < p.width && v < p.height) ? val : 0.0;   // zero fill outside (:548-552)
                            }
                            __builtin_amdgcn_sched_barrier(0);   // at most one pixel row of loads in flight (VGPR pressure)
                        }
                        }
                    }
                }
            }
        }
        const bool active = task >= 0;
        if (!__ballot(active)) break;
        unsigned long long t_now = 0;
        if (p.phase_cycles) { t_now = __builtin_readcyclecounter(); cyc[0] += t_now - t_phase; t_phase = t_now; }

        // ------------------------------------------------------------------ mode A: base sweep
        double best_val = 0.0;      // Select(): bestCoeff = 0.0, index = -1
        int best_idx = -1;
        int best_sel = 0;           // where the winning row lives: ~idx for base, row index in `detail` otherwise
        {
            // Software-pipelined scalar feed: the atom is consumed in four groups of 16 coefficients held in
            // SGPRs.  SMEM returns out of order, so the only usable wait is lgkmcnt(0); therefore exactly one
            // group load is kept in flight: the load of group g+1 is issued right after the wait for group g
            // (i.e. after g's first multiply-add) and has 15 multiply-adds (and the co-resident wave's work)
            // to land.  The row after the last one is a zero pad, so the trailing prefetch stays in bounds.
            scalar_f64_ptr atom = base_s;
            double ga[16], gb[16];
            load_group(ga, atom);
            for (int a = 0; a < p.base_rows_padded; ++a, atom += N) {
                double tot = 0.0;
                MAC_GROUP(ga, gb, 0, atom + 16)
                MAC_GROUP(gb, ga, 16, atom + 32)
                MAC_GROUP(ga, gb, 32, atom + 48)
                MAC_GROUP(gb, ga, 48, atom + 64)
                if (__builtin_fabs(tot) > __builtin_fabs(best_val)) { best_val = tot; best_idx = a; }
            }
            best_sel = ~best_idx;
        }

        if (p.phase_cycles) { t_now = __builtin_readcyclecounter(); cyc[1] += t_now - t_phase; t_phase = t_now; }
        // ------------------------------------------------------------------ mode B: per-tile-channel detail blocks
        {
            const bool has_blocks = active && nblk > 0;
            const unsigned long long has = __ballot(has_blocks);
            if (has) {
#pragma unroll 1
                for (int batch = 0; batch < 64 / STAGE_LANES; ++batch) {
                    unsigned bm = (unsigned)((has >> (STAGE_LANES * batch)) & ((1u << STAGE_LANES) - 1u));
                    if (!bm) continue;
                    __syncthreads();
                    if (has_blocks && (lane / STAGE_LANES) == batch) {
                        const int sl = lane % STAGE_LANES;
#pragma unroll
                        for (int jj = 0; jj < N / 2; ++jj) {
                            stage[jj][sl][0] = r[2 * jj];
                            stage[jj][sl][1] = r[2 * jj + 1];
                        }
                    }
                    __syncthreads();
                    while (bm) {
                        const int sl = __builtin_ctz(bm);
                        bm &= bm - 1;
                        const int t = __builtin_amdgcn_readfirstlane(batch * STAGE_LANES + sl);
                        const int nb = wave_bcast_i32(nblk, t);
                        const int cht = wave_bcast_i32(ch, t);
                        int off = p.num_base;
#pragma unroll 1
                        for (int i = 0; i < nb; ++i) {
                            const unsigned e = blocks[i][t];
                            const int blk = (int)(e & 0x7FFFu);
                            const int rows = rows_s[blk];
                            if (!(e & 0x8000u)) {        // a repeated block can never win (identical projections, higher index)
                                const double2* src = (const double2*)p.detail_t +
                                                     ((long long)cht * p.num_base + blk) * (N / 2) * 64 + lane;
                                double tot = 0.0;
#pragma unroll 8
                                for (int jj = 0; jj < N / 2; ++jj) {
                                    const double2 a = src[jj * 64];
                                    const double r0 = stage[jj][sl][0];
                                    const double r1 = stage[jj][sl][1];
                                    tot += a.x * r0;
                                    tot += a.y * r1;
                                }
                                // argmax over the 64 rows: larger |p| wins, exact ties go to the lower row
                                // (= lower dictionary index, what the reference's first-come strict '>' scan yields);
                                // a NaN gets key -1 and can never win (MatchingPursuit.cpp:16 compares with '>').
                                double key = (tot != tot) ? -1.0 : __builtin_fabs(tot);
                                double val = tot;
                                int row = lane;
#pragma unroll
                                for (int m = 32; m >= 1; m >>= 1) {
                                    const double okey = __shfl_xor(key, m);
                                    const double oval = __shfl_xor(val, m);
                                    const int orow = __shfl_xor(row, m);
                                    const bool take = (okey > key) || (okey == key && orow < row);
                                    key = take ? okey : key;
                                    val = take ? oval : val;
                                    row = take ? orow : row;
                                }
                                if (lane == t && key > __builtin_fabs(best_val)) {
                                    best_val = val;
                                    best_idx = off + row;
                                    best_sel = rowoff_s[blk] + row;
                                }
                            }
                            off += rows;
                        }
                    }
                }
            }
        }

        if (p.phase_cycles) { t_now = __builtin_readcyclecounter(); cyc[2] += t_now - t_phase; t_phase = t_now; }
        // ------------------------------------------------------------------ quantise, record, update (per lane)
        if (active) {
            swept += (unsigned)(p.num_base + extra_rows);
            bool done = false;
            int count = 0;
            unsigned record = 0;
            if (best_idx < 0) {                                   // MatchingPursuit.cpp:50-54
                done = true; count = step; record = 0;
            } else {
                const unsigned delta = (step > 0) ? (((unsigned)(best_idx - prev_id) << 1) ^ (unsigned)((best_idx - prev_id) >> 31))
                                                  : (unsigned)best_idx;
                prev_id = best_idx;
                const double qstep = p.quant[ch * K + step];
                const int q = (int)__builtin_round(best_val / qstep);
                const unsigned zz = ((unsigned)q << 1) ^ (unsigned)(q >> 31);
                record = (delta & 0xFFFFu) | ((zz & 0xFFFFu) << 16);
                if (q == 0) {                                     // :66-69
                    done = true; count = step;
                } else {
                    const double coeff = qstep * (double)q;
                    const double* row = (best_sel < 0) ? (p.base + (long long)(~best_sel) * N)
                                                       : (p.detail + ((long long)ch * p.detail_rows + best_sel) * N);
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const double scaled = coeff * row[j];     // Vector::Scale
                        r[j] = r[j] - scaled;                     // Vector::Subtract
                        if ((j & 15) == 15) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight
                    }
                    if (best_idx < p.num_base) {                  // CompressedImage.cpp:226-229: unlock DetailBasis[choice]
                        unsigned e = (unsigned)best_idx;
                        for (int i = 0; i < nblk; ++i)
                            if ((blocks[i][lane] & 0x7FFFu) == (unsigned)best_idx) e |= 0x8000u;
                        blocks[nblk][lane] = (uint16_t)e;
                        ++nblk;
                        extra_rows += rows_s[best_idx];
                    }
                }
            }
            p.choices[(long long)out_index * K + step] = record;
            if (!done) {
                ++step;
                if (step == K) { done = true; count = K; }
            }
            if (done) {
                double e2 = 0.0;
#pragma unroll
                for (int j = 0; j < N; ++j) e2 += r[j] * r[j];
                p.counts[out_index] = (uint16_t)count;
                p.energy[out_index] = e2;
                p.swept[out_index] = swept;
                task = -1;
            }
        }
        ++iterations;
        if (p.phase_cycles) { t_now = __builtin_readcyclecounter(); cyc[3] += t_now - t_phase; }
    }
    if (p.phase_cycles && lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&p.phase_cycles[i], cyc[i]);
        atomicAdd(&p.phase_cycles[4], (unsigned long long)iterations);
    }
}

// Per-stream symbol histograms for the Huffman/Golomb stage (RCCL all-reduce input):
// stream 0 = lengths, stream 1 + 2K*ch + 2i (+1) = codes[2K*ch + 2i (+1)] (CompressedImage.cpp:556-572).
__global__ __launch_bounds__(256) void mp_histogram_kernel(const HistParams p)
{
    const long long n = p.tiles * 3;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(o % 3);
        const int cnt = p.counts[o];
        atomicAdd(&p.hist[cnt & (kHistBins - 1)], 1u);
        for (int i = 0; i < cnt; ++i) {
            const uint32_t rec = p.choices[o * p.K + i];
            const int s = 1 + 2 * p.K * ch + 2 * i;
            atomicAdd(&p.hist[(long long)s * kHistBins + (rec & 0xFFFFu & (kHistBins - 1))], 1u);
            atomicAdd(&p.hist[(long long)(s + 1) * kHistBins + ((rec >> 16) & (kHistBins - 1))], 1u);
        }
    }
}

int launch_encode(const EncodeParams& p, int waves, void* stream)
{
    if (waves < 1) waves = 1;
    hipLaunchKernelGGL(mp_encode_kernel, dim3((unsigned)waves), dim3(64), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int launch_histogram(const HistParams& p, void* stream)
{
    long long n = p.tiles * 3;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mp_histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

int encode_max_resident_waves()
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1024;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mp_encode_kernel, 64, 0) != hipSuccess || per_cu < 1) per_cu = 4;
    return cus * per_cu;
}

}  // namespace mpc
