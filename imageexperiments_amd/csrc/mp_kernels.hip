// mp_kernels.hip -- product: hand-written gfx950 (CDNA4, wave64) kernels of the quantized
// matching-pursuit tile encoder.  See mp_device.h for the pipeline.  Integer outputs must equal the reference's
// double path (MatchingPursuit.cpp:39-74) bit for bit, so every value that decides anything is IEEE double with
// separately rounded mul/add (-ffp-contract=off) in the reference's order:
//     tot = 0; tot += row[j] * r[j]   (j ascending, mathmatrix.cpp:436-444)
// Two ways to find the row Select() returns:
//   * filtered sweeps (default): split-bfloat16 MFMA approximations of ALL rows with a proven error bound pick the
//     one or two rows per tile-channel that can be the maximum; only those get the dot product above
//     (mp_filter_wave_kernel, mp_detail_filter_kernel; comment block "filtered sweeps").
//   * exhaustive sweeps (MPC_FILTER=0, the product's own cross-check): every row gets it -- r (one tile-channel per
//     lane) lives in 128 VGPRs, the row is wave-uniform and arrives through the scalar data cache (s_load_dwordx16
//     -> SGPR operand of v_mul_f64).  SMEM returns out of order, so the only usable wait is lgkmcnt(0): the loop
//     keeps exactly ONE 16-double group load in flight (mp_base_kernel, mp_detail_kernel; measured:
//     tools/ubench_scalar_sweep.hip).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <type_traits>

#include "mp_device.h"

namespace mpc {

namespace {

typedef const double __attribute__((address_space(4))) * scalar_f64_ptr;   // constant address space: forces s_load
typedef const uint32_t __attribute__((address_space(4))) * scalar_u32_ptr;

constexpr int N = 64;                  // pixels per tile

// YUV weights, ImageHelper/inc/misc.h:7-11; folded in double exactly like the reference's macros
constexpr double W_R = 0.299, W_G = 0.587, W_B = 0.114;
constexpr double U_SCALE = 0.436 / (1.0 - 0.114);
constexpr double V_SCALE = 0.615 / (1.0 - 0.299);

__device__ __forceinline__ void load_group(double (&g)[16], scalar_f64_ptr src) {
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] = src[i];
}

// one 16-term group of the sequential dot product; `cur` = scalar group in SGPRs, `vec` = the VGPR operand
#define MAC_GROUP(cur, nxt, vec, jbase, nextptr)                                         \
    tot += cur[0] * vec[jbase];                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    load_group(nxt, nextptr);                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    _Pragma("unroll") for (int i = 1; i < 16; ++i) tot += cur[i] * vec[(jbase) + i];     \
    __builtin_amdgcn_sched_barrier(0);

// A dictionary-row pointer for the scalar unit.  Both halves go through readfirstlane: the intrinsic's result is
// uniform by definition, so the loads below stay s_load whatever the compiler's divergence analysis concluded
// about the index arithmetic that produced `p` (it has flipped between builds; a vector-load sweep is 6x slower).
__device__ __forceinline__ scalar_f64_ptr as_scalar(const double* p)
{
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (scalar_f64_ptr)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}

}  // namespace

// --------------------------------------------------------------------------------------------------
// init.  Tile mode: one wave per tile, LANE = PIXEL (block index dx + 8*dy): the pixel is read once, converted
// (misc.cpp:12-21) and the three channel residuals are written as three coalesced 512-byte rows; lane 0 resets
// the three tile-channels' pursuit state.  Vector mode (CalcMPDynamic on caller vectors): one wave per vector.
// Block 0 also resets the batch counters.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mp_init_kernel(const Workspace ws, const FrameInput in, long long tc_begin, int n)
{
    const int lane = threadIdx.x;
    const bool vec = in.vec_in != nullptr;
    if (blockIdx.x == 0) {
        for (int b = lane; b < kNumBuckets; b += 64) {
            ws.bucket_count[0][b] = 0;
            ws.bucket_count[1][b] = 0;
            ws.bucket_cursor[b] = 0;
        }
        if (lane < 3) {
            ws.counters[lane] = vec ? (lane == in.vec_channel ? (unsigned)n : 0u) : (unsigned)(n / 3);
            ws.counters[3 + lane] = 0;
            ws.counters[6 + lane] = 0;                       // chunk cursor, chunk count, item count
        }
    }
    const int per_wave = vec ? 1 : 3;                        // tile-channels handled by one wave
    const int work = (n + per_wave - 1) / per_wave;
    for (int w = blockIdx.x; w < work; w += gridDim.x) {
        const int tc0 = w * per_wave;
        const long long gtc0 = tc_begin + tc0;
        if (vec) {
            ws.r[(long long)tc0 * N + lane] = in.vec_in[gtc0 * N + lane];
            if (lane == 0) {
                ws.prev_id[tc0] = 0; ws.nblk[tc0] = 0; ws.extra_rows[tc0] = 0; ws.swept[tc0] = 0;
                ws.out_index[tc0] = (int)((unsigned)gtc0 | ((unsigned)in.vec_channel << 30));
                ws.act[0][in.vec_channel][tc0] = tc0;
            }
            continue;
        }
        const long long unit = gtc0 / 3;                     // batches start on a unit boundary
        const int tiles_per_frame = in.tiles_x * in.tile_rows;
        const int frame = (int)(unit / tiles_per_frame);
        const int tile = (int)(unit - (long long)frame * tiles_per_frame);
        const int tx = tile / in.tile_rows;
        const int ty = in.tile_row_begin + (tile - tx * in.tile_rows);
        const int u = tx * 8 + (lane & 7), v = ty * 8 + (lane >> 3);
        double Y = 0.0, U = 0.0, V = 0.0;                    // zero fill outside (CompressedImage.cpp:548-552)
        if (u < in.width && v < in.height) {
            const uint8_t* px = in.rgb + (long long)frame * in.frame_stride + (long long)v * in.row_stride + 3 * u;
            const double red = (double)px[0], green = (double)px[1], blue = (double)px[2];
            Y = (W_R * red + W_G * green + W_B * blue);
            U = (U_SCALE * (blue - Y));
            V = (V_SCALE * (red - Y));
        }
        ws.r[(long long)(tc0 + 0) * N + lane] = Y;
        ws.r[(long long)(tc0 + 1) * N + lane] = U;
        ws.r[(long long)(tc0 + 2) * N + lane] = V;
        if (lane < 3) {
            const int tc = tc0 + lane;
            ws.prev_id[tc] = 0; ws.nblk[tc] = 0; ws.extra_rows[tc] = 0; ws.swept[tc] = 0;
            ws.out_index[tc] = (int)((unsigned)(gtc0 + lane) | ((unsigned)lane << 30));   // record index | channel << 30
            ws.act[0][lane][tc0 / 3] = tc;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// The sweep: `nrows` consecutive dictionary rows (wave-uniform, scalar-fed, one 16-double group load in
// flight) against the 64 residuals held by the wave.  In-register argmax in row order with the reference's
// strict '>' (first maximum wins; Select(): bestCoeff = 0.0, index = -1, MatchingPursuit.cpp:9-19).
// The row after the last one must be readable (zero pads at the end of `base` and `detail`).
// --------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ void load_residual(double (&r)[N], const double* src_row)
{
    const double2* src = (const double2*)src_row;
#pragma unroll
    for (int jj = 0; jj < N / 2; ++jj) {
        const double2 v = src[jj];
        r[2 * jj] = v.x;
        r[2 * jj + 1] = v.y;
    }
}

__device__ __forceinline__ void sweep_rows(const double (&r)[N], scalar_f64_ptr rows, int nrows, double& best_val,
                                           int& best_row)
{
    best_val = 0.0;
    best_row = -1;
    double ga[16], gb[16];
    load_group(ga, rows);
    for (int a = 0; a < nrows; ++a, rows += N) {
        double tot = 0.0;
        MAC_GROUP(ga, gb, r, 0, rows + 16)
        MAC_GROUP(gb, ga, r, 16, rows + 32)
        MAC_GROUP(ga, gb, r, 32, rows + 48)
        MAC_GROUP(gb, ga, r, 48, rows + 64)
        if (__builtin_fabs(tot) > __builtin_fabs(best_val)) { best_val = tot; best_row = a; }
    }
}

// Block-list entry (u16): bits 0-8 chosen base atom (= detail block), bits 9-14 rows of that block (62 | 63), bit 15
// = repeat of an earlier entry.  Carrying the row count saves the finish kernel a dependent gather per entry.
__device__ __forceinline__ int entry_block(unsigned e) { return (int)(e & 0x1FFu); }
__device__ __forceinline__ int entry_rows(unsigned e) { return (int)((e >> 9) & 0x3Fu); }
__device__ __forceinline__ bool entry_repeat(unsigned e) { return (e & 0x8000u) != 0; }
__device__ __forceinline__ bool entry_is_item(unsigned e) { return !entry_repeat(e) && entry_block(e) != 0; }   // bucketed sweep

// nblk[tc]: entries in blk_list (low byte) | 0x100 once block 0 (DetailBasis[0]) is unlocked
__device__ __forceinline__ int nblk_count(int v) { return v & 0xFF; }
__device__ __forceinline__ bool nblk_has0(int v) { return (v & 0x100) != 0; }

__device__ __forceinline__ unsigned scalar_counter(const unsigned* counters, int i)
{
    return ((scalar_u32_ptr)(uintptr_t)counters)[i];
}

}  // namespace

// Work-unit decoding shared by the sweep kernels: the three per-channel active lists are cut into groups of 64
// entries; unit u of a kernel = (sub-range `part`, group).  Units are numbered part-major so that waves running
// at the same time sweep the same dictionary rows (scalar-cache / L2 locality).  Grids are persistent
// (gridDim = a few waves per SIMD) and stride over the units, whose number is only known on the device.
struct UnitMap {
    int g0, g1, g2;            // groups per channel
    __device__ int groups() const { return g0 + g1 + g2; }
};

__device__ __forceinline__ UnitMap unit_map(const unsigned* counters, int cur)
{
    UnitMap m;
    m.g0 = ((int)scalar_counter(counters, cur * 3 + 0) + 63) >> 6;
    m.g1 = ((int)scalar_counter(counters, cur * 3 + 1) + 63) >> 6;
    m.g2 = ((int)scalar_counter(counters, cur * 3 + 2) + 63) >> 6;
    return m;
}

__device__ __forceinline__ void decode_group(const UnitMap& m, int gi, int& ch, int& group)
{
    int c, g;
    if (gi < m.g0) { c = 0; g = gi; }
    else if (gi < m.g0 + m.g1) { c = 1; g = gi - m.g0; }
    else { c = 2; g = gi - m.g0 - m.g1; }
    // wave-uniform by construction; say so, or the dictionary-row loads that depend on them stop being scalar
    ch = __builtin_amdgcn_readfirstlane(c);
    group = __builtin_amdgcn_readfirstlane(g);
}

// How finely a step's sweeps are cut is decided on the device from the active counts, identically by the
// producer (sweep kernel) and the consumer (finish kernel): enough units for ~4 per wave of the persistent
// grid (tail quantisation <= ~12 %), but no finer -- every extra range re-gathers the residuals and adds a
// candidate for the finish kernel to read.
__device__ __forceinline__ int effective_parts(int work_groups, int grid_waves, int max_parts)
{
    if (work_groups < 1) return 1;
    int p = (4 * grid_waves + work_groups - 1) / work_groups;
    if (p < 1) p = 1;
    // wave-uniform by construction; tell the compiler, or everything derived from it (row pointers!) is
    // treated as divergent and the dictionary loads stop being scalar
    return __builtin_amdgcn_readfirstlane(p > max_parts ? max_parts : p);
}

// base sweep: unit = 64 entries of one channel's active list x one of `parts` ranges of the 510 base atoms;
// the ranges are combined in index order by the finish kernel.  With `with_detail0` one more range per group
// sweeps DetailBasis[0] of the group's channel (63 rows, unlocked by the DC atom, i.e. held by nearly every
// tile-channel after step 0 -- no bucketing needed; lanes that have not unlocked it idle).
__global__ __launch_bounds__(64, 3) void mp_base_kernel(const Workspace ws, const DictDevice dict, int cur, int max_parts,
                                                        int with_detail0, int grid_hint)
{
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && lane < 3) ws.counters[(cur ^ 1) * 3 + lane] = 0;      // next step's active counts
    const UnitMap m = unit_map(ws.counters, cur);
    const int total_groups = m.groups();
    const int parts = effective_parts(total_groups, grid_hint, max_parts);
    const int n_units = total_groups * (parts + with_detail0);
    const int per = __builtin_amdgcn_readfirstlane((dict.base_rows_padded + parts - 1) / parts);
    const int rows0 = __builtin_amdgcn_readfirstlane(dict.block_rows[0]);
    for (int u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int part = __builtin_amdgcn_readfirstlane(u / total_groups);
        int ch, group;
        decode_group(m, u - part * total_groups, ch, group);
        const int n_act = (int)scalar_counter(ws.counters, cur * 3 + ch);
        const int pos = group * 64 + lane;
        const bool valid = pos < n_act;
        const int tc = ws.act[cur][ch][valid ? pos : group * 64];
        if (part < parts) {
            double r[N];
            load_residual(r, ws.r + (long long)tc * N);
            const int a0 = part * per;
            const int a1 = (a0 + per < dict.base_rows_padded) ? a0 + per : dict.base_rows_padded;
            double best_val;
            int best_row;
            sweep_rows(r, as_scalar(dict.base + (long long)a0 * N), __builtin_amdgcn_readfirstlane(a1 - a0), best_val, best_row);
            if (valid) {
                ws.part_val[(long long)tc * kMaxParts + part] = best_val;
                ws.part_idx[(long long)tc * kMaxParts + part] = (best_row < 0) ? -1 : a0 + best_row;
            }
        } else {
            bool has0 = false;
            if (valid) {
                has0 = nblk_has0(ws.nblk[tc]);
            }
            if (!__ballot(has0)) continue;
            double r[N];
            load_residual(r, ws.r + (long long)tc * N);
            double best_val;
            int best_row;
            sweep_rows(r, as_scalar(dict.detail + (long long)ch * dict.detail_rows * N), rows0, best_val, best_row);
            if (has0) {
                ws.cand0_val[tc] = best_val;
                ws.cand0_row[tc] = best_row;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------
// filtered sweeps.  The argmax of Select() only needs the exact projection of the rows that can be the
// maximum.  So: (1) approximate ALL projections of a group of 16 tile-channels in split bfloat16 on the matrix
// cores, (2) keep, per tile-channel, the rows whose approximation is within 2E of the largest one, (3) evaluate
// only those in the reference's arithmetic (sequential double dot product, mathmatrix.cpp:436-444) and pick the
// first strict maximum among them in row order.  The result is bit-identical to sweeping every row:
//   * the approximation is a split-bfloat16 product on v_mfma_f32_16x16x32_bf16: residual and row elements are
//     x = hi + lo + e, hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-16 |x| (u = 2^-8 per rounding), and the three MFMAs
//     hi*hi + hi*lo + lo*hi leave out lo*lo <= 2^-16 |r_j b_j|: at most 3 * 2^-16 * sum|r_j b_j| in all.  The products
//     of two 8-bit significands are exact in f32; accumulating 6 x 32 of them in f32 (any order, any rounding mode)
//     adds at most 6 * 33 * 2^-23 of sum|r_j b_j|.  With sum|r_j b_j| <= |r|_2 |b|_2 and |b|_2 <= 1 + 2^-50 that is
//     < 7.0e-5 |r|_2, plus < 2^-110 where operands or products are subnormal; the reference's double value differs
//     from the real dot product by < 2^-46 |r|_2.  E = 2^-13 |r~|_2 + 2^-100 (r~ = the f32-rounded residual) leaves a
//     factor ~1.7 over all of it.
//   * let j* be the row Select() returns (lowest index with maximal exact |p|) and j~ the approximate maximum:
//     approx|p_j*| >= |p_j*| - E >= |p_j~| - E >= approx|p_j~| - 2E, so j* is kept -- as is every row tying with it.
//   * overflow / NaN anywhere makes the comparison `approx < threshold` false: the row is kept and evaluated.
//   * the threshold may use the largest approximation over ANY subset of the rows the tile-channel can choose from:
//     a smaller maximum only lowers it.  Base rows and block 0 share one (mp_filter_wave_kernel), which it also leaves in
//     approx_max[] for the detail blocks (mp_detail_filter_kernel): a block whose best row is far below the base
//     maximum yields no survivor at all and reports "none" (index -1), which the finish kernel skips.
//   * rows that are +-copies of an earlier row (base row 509 = -row 0) are zero in the filter copy: they tie with the
//     earlier row exactly and can never be returned (host_dictionary.cpp: filter_tiles).
// --------------------------------------------------------------------------------------------------
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFilterGroup = 16;                                          // tile-channels per MFMA column block
constexpr int kStageStride = N + 1;                                       // doubles per staged residual (+1: bank spread)
constexpr float kFilterSlack = 0x1p-13f;
constexpr float kFilterAbs = 0x1p-100f;

typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned bf16_bits(float x)       // round to nearest even, like host_dictionary.cpp: bf16_round
{
    const unsigned bits = __float_as_uint(x);
    return (bits + 0x7FFFu + ((bits >> 16) & 1u)) >> 16;
}

// B operand of the 16x16x32 MFMAs for 16 staged residuals: lane (slot = l & 15, h = l >> 4) holds, for kk = 0, 1, the
// eight elements r~[slot][32kk + 8h + j] split into hi[kk] and lo[kk]; also returns |r~[slot]|^2 (summed over the
// four h lanes)
__device__ __forceinline__ double load_b_operand(bf16x8 (&hi)[2], bf16x8 (&lo)[2], const double* stage, int lane)
{
    const double* src = stage + (lane & 15) * kStageStride + 8 * (lane >> 4);
    double ss = 0.0;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (float)src[32 * kk + j];
            ss += (double)x * (double)x;
            const unsigned h = bf16_bits(x);
            const unsigned l = bf16_bits(x - __uint_as_float(h << 16));
            hi[kk][j] = (short)h;
            lo[kk][j] = (short)l;
        }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    return ss;
}

// one 16-row tile of the filter copy (four 16-byte operands per lane: hi/lo x two k halves) against the B operand
__device__ __forceinline__ f32x4 filter_tile_mfma(const uint4 (&a)[4], const bf16x8 (&hi)[2], const bf16x8 (&lo)[2])
{
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        bf16x8 ah, al;
        __builtin_memcpy(&ah, &a[2 * kk + 0], 16);
        __builtin_memcpy(&al, &a[2 * kk + 1], 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, hi[kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, lo[kk], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, hi[kk], acc, 0, 0, 0);
    }
    return acc;
}

// mask of the lowest n bits (n may be <= 0 or >= 32)
__device__ __forceinline__ unsigned low_bits(int n) { return n <= 0 ? 0u : (n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u); }

// keep the better of two (value, row) results of disjoint row sets: larger |value|, on a tie the lower row
__device__ __forceinline__ void keep_better(double& v, int& i, double ov, int oi)
{
    if (oi >= 0 && (i < 0 || __builtin_fabs(ov) > __builtin_fabs(v) || (__builtin_fabs(ov) == __builtin_fabs(v) && oi < i))) {
        v = ov;
        i = oi;
    }
}

// Exact evaluation of the surviving rows, cooperatively by one wave.  Lane (slot, sub) holds its survivors as bit
// masks (bit i of segment s = row off_s + 4i + sub of rows_s).  Per round the first 16 lanes that still have one
// each put up their lowest survivor; the wave then works lane = pixel: one coalesced 512-byte read per row, the
// 64 products row[j]*r[j] (rounded to double like the reference's `l*r`) go to LDS, and lane k adds up the
// products of survivor k in j order -- exactly the reference's `tot += l*r` chain.  A lane's survivors come up in
// ascending row order, so its strict '>' keeps the first maximum.
constexpr int kCandRound = 16;
struct CandLds {
    double prod[kCandRound * kStageStride];
    double res[kCandRound];
};

// LDS hand-off between lanes of ONE wave: DS operations of a wave execute in issue order, so all that is needed is
// that the compiler keeps the order and the data has landed; no vector-memory wait (a fence would also drain the
// prefetched global loads)
__device__ __forceinline__ void wave_lds_sync()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void evaluate_survivors(CandLds& lds, const double* stage, int lane, unsigned mask0, unsigned mask1,
                                                   int off0, int off1, const double* rows0, const double* rows1, double& v0,
                                                   int& i0, double& v1, int& i1)
{
    const int slot = lane & 15, sub = lane >> 4;
    v0 = 0.0; v1 = 0.0;
    i0 = -1; i1 = -1;
    for (;;) {
        const bool pending = (mask0 | mask1) != 0;
        const unsigned long long votes = __ballot(pending);
        if (!votes) break;
        const int rank = __popcll(votes & ((1ULL << lane) - 1ULL));
        const bool active = pending && rank < kCandRound;
        int seg = 0, a = 0;
        unsigned long long row_addr = 0;
        if (active) {
            if (mask0) { const int i = __builtin_ctz(mask0); mask0 &= mask0 - 1; a = off0 + 4 * i + sub; }
            else { const int i = __builtin_ctz(mask1); mask1 &= mask1 - 1; a = off1 + 4 * i + sub; seg = 1; }
            row_addr = (unsigned long long)(uintptr_t)((seg ? rows1 : rows0) + (long long)a * N);
        }
        int n = __popcll(votes);
        if (n > kCandRound) n = kCandRound;
        unsigned long long left = votes;
        for (int k0 = 0; k0 < n; k0 += 4) {                     // four row reads in flight; slots past n repeat the last row
            double x[4], y[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int src = __builtin_ctzll(left);         // wave-uniform: the lane that put up survivor k0 + kk
                if (left & (left - 1)) left &= left - 1;
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)row_addr, src);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(row_addr >> 32), src);
                const double* row = (const double*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
                x[kk] = row[lane];
                y[kk] = stage[(src & 15) * kStageStride + lane];
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) lds.prod[(k0 + kk) * kStageStride + lane] = x[kk] * y[kk];
        }
        wave_lds_sync();
        if (lane < n) {
            const double* pr = lds.prod + lane * kStageStride;
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) tot += pr[j];
            lds.res[lane] = tot;
        }
        wave_lds_sync();
        if (active) {
            const double p = lds.res[rank];
            if (seg == 0) { if (__builtin_fabs(p) > __builtin_fabs(v0)) { v0 = p; i0 = a; } }
            else if (__builtin_fabs(p) > __builtin_fabs(v1)) { v1 = p; i1 = a; }
        }
        wave_lds_sync();
    }
    (void)slot;
}

}  // namespace

// What the filter kernels touch, and nothing else: passing the whole Workspace / DictDevice keeps ~100 SGPRs of
// pointers alive and spills them into VGPR lanes.
struct FilterArgs {
    const unsigned* counters_in;     // ws.counters
    unsigned* counters_out;          // same array (next step's active counts are reset here)
    const int* act[3];               // this step's active lists
    double* r;
    const double* upd_coeff;         // the previous step's pending residual update (0 = none): r -= coeff * atom
    const int* upd_sel;              // ~index of a base atom, row of `detail` (this channel) otherwise
    const int* nblk;
    double* part_val;
    int* part_idx;
    double* cand0_val;
    int* cand0_row;
    float* approx_max;               // out: largest approximation over base + block 0, per tile-channel
    const double* base;              // dictionary, double
    const double* block0[3];         // DetailBasis[0] of each channel, double
    const uint16_t* base_f32;        // filter copies
    const uint16_t* block0_f32[3];
    int num_base, rows0;
};

struct DetailFilterArgs {
    const unsigned* counters;
    const int* chunks;
    const int* items;
    const double* r;
    double* cand_val;
    int* cand_row;
    const float* approx_max;         // written by mp_filter_wave_kernel earlier in the step
    const double* detail;            // [3][detail_rows][64]
    long long detail_rows;
    const uint16_t* detail_f32;
    const int32_t* block_rows;
    const int32_t* block_row_off;
    int num_base;
};

// --------------------------------------------------------------------------------------------------
// Wave-autonomous filter kernels.  With the approximations on the bf16 matrix cores the MFMAs are cheap, and what
// is left of a filtered sweep is memory latency and synchronisation.  So one WAVE takes up to 64 tile-channels (four
// column groups of 16) and keeps everything in registers: the B operands are read straight from the residual rows,
// pass 1 runs all row tiles through the MFMAs and only keeps each lane's running maximum, pass 2 runs them again and
// compares against the threshold, queueing the few survivors for the cooperative exact evaluation.  No workgroup
// barrier, no LDS besides the 8 KiB product buffer; each dictionary tile read serves up to 64 tile-channels.
// --------------------------------------------------------------------------------------------------
namespace {

// B operand of one column group read from global memory: this lane's slot row, elements 32kk + 8h + j
__device__ __forceinline__ double load_b_global(bf16x8 (&hi)[2], bf16x8 (&lo)[2], double* row, const double* atom, double coeff,
                                                int lane, bool& nonzero)
{
    double2* dst = (double2*)(row + 8 * (lane >> 4));
    double2 v[8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) v[4 * kk + j2] = dst[16 * kk + j2];
    if (coeff != 0.0) {
        // the residual update the finish kernel decided on (Vector::Scale then Vector::Subtract, mathvector.cpp:116-148:
        // two roundings), applied to this lane's 16 elements on their way in and written back for the detail sweep,
        // the exact evaluations and the next step
        const double2* src = (const double2*)(atom + 8 * (lane >> 4));
        double2 a[8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) a[4 * kk + j2] = src[16 * kk + j2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double sx = coeff * a[i].x, sy = coeff * a[i].y;
            v[i].x = v[i].x - sx;
            v[i].y = v[i].y - sy;
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) dst[16 * kk + j2] = v[4 * kk + j2];
    }
    double ss = 0.0;
    bool nzl = false;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double d = (j & 1) ? v[4 * kk + (j >> 1)].y : v[4 * kk + (j >> 1)].x;
            nzl = nzl || (d != 0.0);
            const float x = (float)d;
            ss += (double)x * (double)x;
            const unsigned h = bf16_bits(x);
            const unsigned l = bf16_bits(x - __uint_as_float(h << 16));
            hi[kk][j] = (short)h;
            lo[kk][j] = (short)l;
        }
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    int nzi = nzl ? 1 : 0;
    nzi |= __shfl_xor(nzi, 16);
    nzi |= __shfl_xor(nzi, 32);
    nonzero = nzi != 0;
    return ss;
}

// The two largest approximations a lane has seen among its rows of one segment, and the row of the largest: if the
// runner-up of the whole tile-channel stays below the threshold, the single survivor is known after pass 1 and the
// second MFMA pass is not needed for it.
struct TopTwo {
    float m1 = 0.0f, m2 = 0.0f;                        // m1 >= m2 throughout
    float sum = 0.0f;                                  // of every |value| seen: a NaN or an infinity poisons it
    int row = -1;
    // five VALU instructions per value (this runs 4 x 576 times per tile-channel group and step): v_cmp + v_cndmask for
    // the row, v_med3 (the middle of {m1, m2, a} is the new runner-up), v_max, v_add
    __device__ __forceinline__ void see(float value, int r)
    {
        const float a = fabsf(value);
        row = a > m1 ? r : row;
        m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
        m1 = fmaxf(m1, a);
        sum += a;
    }
    // fold in the three other lanes that hold rows of the same tile-channel: top = its largest approximation, second =
    // its runner-up (a tie for the top counts as a runner-up), mine = this lane owns the unique top row
    __device__ __forceinline__ void across_lanes(float& top, float& second, bool& mine, bool& any_odd) const
    {
        top = fmaxf(m1, __shfl_xor(m1, 16));
        top = fmaxf(top, __shfl_xor(top, 32));
        const bool at_top = m1 == top;
        int n_top = at_top ? 1 : 0;
        n_top += __shfl_xor(n_top, 16);
        n_top += __shfl_xor(n_top, 32);
        float rest = at_top ? m2 : m1;
        rest = fmaxf(rest, __shfl_xor(rest, 16));
        rest = fmaxf(rest, __shfl_xor(rest, 32));
        second = n_top > 1 ? top : rest;
        mine = at_top && n_top == 1;
        int o = !(sum <= 3.4028234663852886e38f) ? 1 : 0;          // saw a NaN or an infinity (or overflowed): leave it to pass 2
        o |= __shfl_xor(o, 16);
        o |= __shfl_xor(o, 32);
        any_odd = o != 0;
    }
};

// Per-lane FIFO of surviving rows waiting for their exact evaluation.  code = group << 12 | segment << 11 | row.
struct SurvivorQueue {
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
    int count = 0;
    static constexpr int kCapacity = 6;
    __device__ __forceinline__ void push(int code)
    {
        if (count == 0) q0 = code; else if (count == 1) q1 = code; else if (count == 2) q2 = code;
        else if (count == 3) q3 = code; else if (count == 4) q4 = code; else q5 = code;
        ++count;
    }
    __device__ __forceinline__ int pop()
    {
        const int head = q0;
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5;
        --count;
        return head;
    }
};

// Best exact projection seen so far by this lane, per column group and segment (0 = base rows | the block, 1 = block 0)
struct LaneBest {
    double v[4][2];
    int i[4][2];
};

// Drain the queues: rounds of up to 16 survivors (one per lane that has any), evaluated lane = pixel as in
// evaluate_survivors; `res0..3` are this lane's residual rows for the four column groups.
__device__ __forceinline__ void drain_survivors(CandLds& lds, int lane, SurvivorQueue& q, const double* rows_seg0,
                                                const double* rows_seg1, const double* res0, const double* res1,
                                                const double* res2, const double* res3, LaneBest& best)
{
    for (;;) {
        const bool pending = q.count > 0;
        const unsigned long long votes = __ballot(pending);
        if (!votes) break;
        const int rank = __popcll(votes & ((1ULL << lane) - 1ULL));
        const bool active = pending && rank < kCandRound;
        int code = 0;
        unsigned long long row_addr = 0, res_addr = 0;
        if (active) {
            code = q.pop();
            const int g = code >> 12;
            row_addr = (unsigned long long)(uintptr_t)(((code >> 11) & 1 ? rows_seg1 : rows_seg0) + (long long)(code & 0x7FF) * N);
            res_addr = (unsigned long long)(uintptr_t)(g == 0 ? res0 : (g == 1 ? res1 : (g == 2 ? res2 : res3)));
        }
        int n = __popcll(votes);
        if (n > kCandRound) n = kCandRound;
        unsigned long long left = votes;
        for (int k0 = 0; k0 < n; k0 += 8) {                     // eight survivors' reads in flight; slots past n repeat the last
            double x[8], y[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int src = __builtin_ctzll(left);         // wave-uniform: the lane that put up survivor k0 + kk
                if (left & (left - 1)) left &= left - 1;
                const unsigned rl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)row_addr, src);
                const unsigned rh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(row_addr >> 32), src);
                const unsigned sl = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)res_addr, src);
                const unsigned sh = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(res_addr >> 32), src);
                x[kk] = ((const double*)(uintptr_t)(((unsigned long long)rh << 32) | rl))[lane];
                y[kk] = ((const double*)(uintptr_t)(((unsigned long long)sh << 32) | sl))[lane];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) lds.prod[(k0 + kk) * kStageStride + lane] = x[kk] * y[kk];
        }
        wave_lds_sync();
        if (lane < n) {
            const double* pr = lds.prod + lane * kStageStride;
            double tot = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) tot += pr[j];
            lds.res[lane] = tot;
        }
        wave_lds_sync();
        if (active) {
            const double p = lds.res[rank];
            const int g = code >> 12, seg = (code >> 11) & 1, row = code & 0x7FF;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss)
                    if (g == gg && seg == ss && __builtin_fabs(p) > __builtin_fabs(best.v[gg][ss])) {
                        best.v[gg][ss] = p;
                        best.i[gg][ss] = row;
                    }
        }
        wave_lds_sync();
    }
}

// f(integral_constant<0>), ..., f(integral_constant<COUNT-1>): a loop the optimiser cannot decline to unroll (register
// arrays indexed by a loop variable that survives as a variable end up in scratch memory)
template <int COUNT, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (COUNT > 0) {
        static_for<COUNT - 1>(f);
        f(std::integral_constant<int, COUNT - 1>{});
    }
}

// Run body(tile, operands) over tiles first..last-1 with DEPTH tile reads (4 x 16 B per lane each) in flight ahead of
// the MFMAs.  DEPTH 1 for large batches (four column groups per wave keep the matrix cores busy and the registers
// full); DEPTH 4 for small ones, where a tile's MFMAs are far shorter than its L2 round trip.  last - first is a
// multiple of 4.
template <int DEPTH, class Ptr, class Body>
__device__ __forceinline__ void for_each_tile(int first, int last, Ptr&& tile_ptr, Body&& body)
{
#define MPC_LOAD_TILE(dst_, tile_)                                                                             \
    {                                                                                                          \
        const uint4* p_ = tile_ptr((tile_) < last ? (tile_) : last - 1);                                       \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) dst_[q_] = p_[q_ * 64];                               \
    }
    if constexpr (DEPTH == 1) {
        uint4 a[4], nxt[4];
        MPC_LOAD_TILE(a, first)
        for (int tile = first; tile < last; ++tile) {
            MPC_LOAD_TILE(nxt, tile + 1)
            body(tile, a);
#pragma unroll
            for (int q = 0; q < 4; ++q) a[q] = nxt[q];
        }
    } else {
        static_assert(DEPTH == 4, "ring of four named buffers");
        uint4 r0[4], r1[4], r2[4], r3[4];
        MPC_LOAD_TILE(r0, first + 0)
        MPC_LOAD_TILE(r1, first + 1)
        MPC_LOAD_TILE(r2, first + 2)
        MPC_LOAD_TILE(r3, first + 3)
        for (int t0 = first; t0 < last; t0 += 4) {
            body(t0 + 0, r0);
            MPC_LOAD_TILE(r0, t0 + 4)
            body(t0 + 1, r1);
            MPC_LOAD_TILE(r1, t0 + 5)
            body(t0 + 2, r2);
            MPC_LOAD_TILE(r2, t0 + 6)
            body(t0 + 3, r3);
            MPC_LOAD_TILE(r3, t0 + 7)
        }
    }
#undef MPC_LOAD_TILE
}

}  // namespace

// Base rows and (steps > 0) DetailBasis[0] for up to 64 tile-channels of one channel's active list per wave.
#ifndef MPC_WAVE_OCC
#define MPC_WAVE_OCC 2
#endif
template <int MAXG, int DEPTH>
__global__ __launch_bounds__(64, MPC_WAVE_OCC) void mp_filter_wave_kernel(const FilterArgs fa, int cur, int with_detail0, int force_groups)
{
    __shared__ CandLds s_cand;
    const int lane = threadIdx.x;
    const int slot = lane & 15, sub = lane >> 4;
    if (blockIdx.x == 0 && lane < 3) fa.counters_out[(cur ^ 1) * 3 + lane] = 0;      // next step's active counts
    const int n0 = (int)scalar_counter(fa.counters_in, cur * 3 + 0), n1 = (int)scalar_counter(fa.counters_in, cur * 3 + 1),
              n2 = (int)scalar_counter(fa.counters_in, cur * 3 + 2);
    // column groups per wave: as many as it takes to give every wave of the grid about one unit (small batches are
    // latency-bound: spread them thin; large ones amortise each dictionary tile read over 64 tile-channels)
    int groups = force_groups > 0 ? force_groups : (n0 + n1 + n2 + 16 * (int)gridDim.x - 1) / (16 * (int)gridDim.x);
    groups = __builtin_amdgcn_readfirstlane(groups < 1 ? 1 : (groups > MAXG ? MAXG : groups));
    const int per_unit = 16 * groups;
    const int g0 = (n0 + per_unit - 1) / per_unit, g1 = (n1 + per_unit - 1) / per_unit, g2 = (n2 + per_unit - 1) / per_unit;
    const int ntiles = kBaseFilterTiles + (with_detail0 ? kBlockFilterTiles : 0);
    for (int u = blockIdx.x; u < g0 + g1 + g2; u += gridDim.x) {
        const int ch = __builtin_amdgcn_readfirstlane(u < g0 ? 0 : (u < g0 + g1 ? 1 : 2));
        const int first = (u - (ch == 0 ? 0 : (ch == 1 ? g0 : g0 + g1))) * per_unit;
        const int n_act = ch == 0 ? n0 : (ch == 1 ? n1 : n2);
        const int* act = ch == 0 ? fa.act[0] : (ch == 1 ? fa.act[1] : fa.act[2]);
        const uint16_t* tiles0 = ch == 0 ? fa.block0_f32[0] : (ch == 1 ? fa.block0_f32[1] : fa.block0_f32[2]);
        const double* block0 = ch == 0 ? fa.block0[0] : (ch == 1 ? fa.block0[1] : fa.block0[2]);

        int tc[4] = {0, 0, 0, 0};
        bool ok[4] = {false, false, false, false};
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            const int pos = first + 16 * g + slot;
            ok[g] = g < groups && pos < n_act;
            tc[g] = act[ok[g] ? pos : first];
        }
        bf16x8 bh[4][2], bl[4][2];
        float window[4];
        bool live[4], has0[4];
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            window[g] = 0.0f; live[g] = false; has0[g] = false;
            if (g < groups) {
                bool nz;
                const double coeff = (with_detail0 && ok[g]) ? fa.upd_coeff[tc[g]] : 0.0;       // steps > 0 only
                const int sel = (with_detail0 && ok[g]) ? fa.upd_sel[tc[g]] : 0;
                const double* atom = sel < 0 ? fa.base + (long long)(~sel) * N : block0 + (long long)sel * N;
                const double ss = load_b_global(bh[g], bl[g], fa.r + (long long)tc[g] * N, atom, coeff, lane, nz);
                window[g] = 2.0f * (kFilterSlack * (float)__builtin_sqrt(ss) + kFilterAbs);
                live[g] = ok[g] && nz;                          // an all-zero residual projects to 0 everywhere: index -1
                has0[g] = with_detail0 && ok[g] && nblk_has0(fa.nblk[tc[g]]);
            }
        }
        __threadfence_block();                                  // the updated residuals are read back lane = pixel below
        auto tile_ptr = [&](int tile) {
            return (const uint4*)(tile < kBaseFilterTiles ? fa.base_f32 + tile * 2048 : tiles0 + (tile - kBaseFilterTiles) * 2048) + lane;
        };

        // ---- pass 1: the two largest approximations per lane (its rows: 16*tile + 4*sub + v), base rows and block 0 apart
        TopTwo tb[4], td[4];
        for_each_tile<DEPTH>(0, kBaseFilterTiles, tile_ptr, [&](int tile, const uint4 (&a)[4]) {
            const int row0 = tile * 16 + sub * 4;
            static_for<MAXG>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (g < groups) {
                    const f32x4 acc = filter_tile_mfma(a, bh[g], bl[g]);
#pragma unroll
                    for (int v = 0; v < 4; ++v) tb[g].see(acc[v], row0 + v);
                }
            });
        });
        if (with_detail0)
            for_each_tile<DEPTH>(kBaseFilterTiles, kBaseFilterTiles + kBlockFilterTiles, tile_ptr, [&](int tile, const uint4 (&a)[4]) {
                const int row0 = (tile - kBaseFilterTiles) * 16 + sub * 4;
                static_for<MAXG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if (g < groups) {
                        const f32x4 acc = filter_tile_mfma(a, bh[g], bl[g]);
#pragma unroll
                        for (int v = 0; v < 4; ++v) td[g].see(acc[v], row0 + v);
                    }
                });
            });
        // thresholds; where the runner-up is below it the survivors are known already (at most one per segment)
        LaneBest best;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) { best.v[g][sg] = 0.0; best.i[g][sg] = -1; }
        const double* res0 = fa.r + (long long)tc[0] * N;
        const double* res1 = fa.r + (long long)tc[1] * N;
        const double* res2 = fa.r + (long long)tc[2] * N;
        const double* res3 = fa.r + (long long)tc[3] * N;
        float thr[4];
        bool second_pass[4] = {false, false, false, false};    // wave-uniform per column group
        SurvivorQueue queue;
        static_for<MAXG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            float top_b, second_b, top_d, second_d;
            bool mine_b, mine_d, odd_b, odd_d;
            tb[g].across_lanes(top_b, second_b, mine_b, odd_b);
            td[g].across_lanes(top_d, second_d, mine_d, odd_d);
            const float top = has0[g] ? fmaxf(top_b, top_d) : top_b;      // one threshold for base rows and block 0
            thr[g] = top - window[g];
            if (sub == 0 && ok[g]) fa.approx_max[tc[g]] = top;
            // ambiguous: a runner-up at or above the threshold, a non-positive threshold (zero pads would qualify), NaN/Inf
            const bool unclear = !(second_b < thr[g]) || (has0[g] && !(second_d < thr[g])) || !(thr[g] > 0.0f) || odd_b ||
                                 (has0[g] && odd_d);
            second_pass[g] = g < groups && __ballot(live[g] && unclear) != 0;
            if (g < groups && !second_pass[g] && live[g]) {
                if (mine_b && !(top_b < thr[g])) queue.push((g << 12) | tb[g].row);
                if (has0[g] && mine_d && !(top_d < thr[g])) queue.push((g << 12) | 0x800 | td[g].row);
            }
            if (__ballot(queue.count > SurvivorQueue::kCapacity - 2))       // room for the next group's two
                drain_survivors(s_cand, lane, queue, fa.base, block0, res0, res1, res2, res3, best);
        });

        // ---- pass 2, only for column groups with an unclear tile-channel: the same MFMAs again; rows at or above the
        //      threshold queue up for exact evaluation
        // pass 2 pushes up to four rows of a lane per tile before it looks at the fill level: enter it with room for them
        if ((second_pass[0] || second_pass[1] || second_pass[2] || second_pass[3]) &&
            __ballot(queue.count > SurvivorQueue::kCapacity - 4))
            drain_survivors(s_cand, lane, queue, fa.base, block0, res0, res1, res2, res3, best);
        if (second_pass[0] || second_pass[1] || second_pass[2] || second_pass[3])
            for_each_tile<DEPTH>(0, ntiles, tile_ptr, [&](int tile, const uint4 (&a)[4]) {
                const bool is_base = tile < kBaseFilterTiles;
                const int row0 = (is_base ? tile * 16 : (tile - kBaseFilterTiles) * 16) + sub * 4;
                const int limit = is_base ? fa.num_base : fa.rows0;
                static_for<MAXG>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if (second_pass[g]) {
                        const f32x4 acc = filter_tile_mfma(a, bh[g], bl[g]);
                        const bool wanted = live[g] && (is_base || has0[g]);
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            if (wanted && !(fabsf(acc[v]) < thr[g]) && row0 + v < limit)
                                queue.push((g << 12) | (is_base ? 0 : 0x800) | (row0 + v));
                        if (__ballot(queue.count > SurvivorQueue::kCapacity - 4))
                            drain_survivors(s_cand, lane, queue, fa.base, block0, res0, res1, res2, res3, best);
                    }
                });
            });
        drain_survivors(s_cand, lane, queue, fa.base, block0, res0, res1, res2, res3, best);

        // ---- the four lanes of a slot hold interleaved rows: combine by (|value|, lower row), lane sub == 0 reports
#pragma unroll
        for (int g = 0; g < MAXG; ++g) {
            if (g >= groups) continue;
#pragma unroll
            for (int sg = 0; sg < 2; ++sg) {
                keep_better(best.v[g][sg], best.i[g][sg], __shfl_xor(best.v[g][sg], 16), __shfl_xor(best.i[g][sg], 16));
                keep_better(best.v[g][sg], best.i[g][sg], __shfl_xor(best.v[g][sg], 32), __shfl_xor(best.i[g][sg], 32));
            }
            if (sub == 0 && ok[g]) {
                fa.part_val[(long long)tc[g] * kMaxParts] = best.v[g][0];
                fa.part_idx[(long long)tc[g] * kMaxParts] = best.i[g][0];
                if (has0[g]) {
                    fa.cand0_val[tc[g]] = best.v[g][1];
                    fa.cand0_row[tc[g]] = best.i[g][1];
                }
            }
        }
    }
}

// Row range `part` of `row_parts` of a block with `rows` rows: [lo, hi)
__device__ __forceinline__ void row_range(int rows, int row_parts, int part, int& lo, int& hi)
{
    const int per = __builtin_amdgcn_readfirstlane((rows + row_parts - 1) / row_parts);
    lo = part * per;
    hi = (lo + per < rows) ? lo + per : rows;
    if (lo > rows) lo = rows;
}

// Pull a run of dictionary rows towards this XCD's L2 with four vector loads (one lane per 128-byte line of a
// 32 KiB block) so that the scalar loads of the sweep, which can only keep one 128-byte group in flight, find
// them there.  The loaded values are returned to be kept alive by the caller (summed into a dead store guard).
__device__ __forceinline__ double touch_rows(const double* first_row, int nrows, int lane)
{
    double acc = 0.0;
    const int lines = nrows * 4;                       // 512 B per row = 4 lines
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int line = k * 64 + lane;
        if (line < lines) acc += first_row[line * 16];
    }
    return acc;
}

// every other block: work unit = (chunk of <= 64 items of one (channel, block) bucket) x (row range);
// lane = item.
__global__ __launch_bounds__(64, 3) void mp_detail_kernel(const Workspace ws, const DictDevice dict, int max_row_parts,
                                                          int grid_hint, double* touch_sink)
{
    const int lane = threadIdx.x;
    const int row_parts = effective_parts((int)scalar_counter(ws.counters, 7), grid_hint, max_row_parts);
    const unsigned n_units = scalar_counter(ws.counters, 7) * (unsigned)row_parts;
    double keep = 0.0;
    // static grid-stride assignment: units cost about the same, and a shared dequeue counter saturates at
    // ~88 dequeues/us on this chip -- more than the sweeps themselves for thousands of small units
    for (unsigned u = blockIdx.x; u < n_units; u += gridDim.x) {
        const unsigned n_chunks = n_units / (unsigned)row_parts;
        const int part = __builtin_amdgcn_readfirstlane((int)(u / n_chunks));
        const unsigned c = u - (unsigned)part * n_chunks;
        const int* desc = ws.chunks + 4 * (long long)c;
        const int bucket = __builtin_amdgcn_readfirstlane(desc[0]);
        const int begin = __builtin_amdgcn_readfirstlane(desc[1]);
        const int end = __builtin_amdgcn_readfirstlane(desc[2]);
        const int ch = bucket >> 9, blk = bucket & 511;
        const int cnt = end - begin;
        const bool valid = lane < cnt;
        int lo, hi;
        row_range(__builtin_amdgcn_readfirstlane(dict.block_rows[blk]), row_parts, part, lo, hi);
        const long long first = (long long)ch * dict.detail_rows + __builtin_amdgcn_readfirstlane(dict.block_row_off[blk]) + lo;
        keep += touch_rows(dict.detail + first * N, hi - lo, lane);
        const int tc = ws.items[begin + (valid ? lane : 0)];
        double r[N];
        load_residual(r, ws.r + (long long)tc * N);
        double best_val;
        int best_row;
        sweep_rows(r, as_scalar(dict.detail + first * N), __builtin_amdgcn_readfirstlane(hi - lo), best_val, best_row);
        if (valid) {
            ws.cand_val[(long long)(begin + lane) * kMaxRowParts + part] = best_val;
            ws.cand_row[(long long)(begin + lane) * kMaxRowParts + part] = (best_row < 0) ? -1 : lo + best_row;
        }
    }
    if (keep == 123456.789) touch_sink[lane] = keep;          // never true: keeps the touch loads alive
}

// filtered detail sweep: the same filter for the bucketed detail blocks.  One wave per 16 items of a chunk
// (<= 64 items of one (channel, block) bucket): 64 x 16 approximations (4 tiles, 24 MFMAs), then lane (slot, sub)
// scans every fourth row of its item and the survivors are evaluated exactly.  16 KiB of LDS per wave.
__global__ __launch_bounds__(64) void mp_detail_filter_kernel(const DetailFilterArgs da)
{
    __shared__ __attribute__((aligned(16))) char s_buf[sizeof(CandLds)];   // approximations [row][slot], then the products
    static_assert(sizeof(CandLds) >= sizeof(float) * 64 * kFilterGroup, "s_p fits");
    __shared__ double s_r[kFilterGroup * kStageStride];
    float* s_p = reinterpret_cast<float*>(s_buf);
    CandLds& s_cand = *reinterpret_cast<CandLds*>(s_buf);
    const int lane = threadIdx.x;
    const int slot = lane & 15, sub = lane >> 4;
    const unsigned n_units = scalar_counter(da.counters, 7) * 4u;
    // Workgroups are dealt to the eight XCDs round-robin, and every XCD has an L2 of its own: keep runs of kRun
    // consecutive units -- the quarters of a chunk and the next chunks of the same bucket, which read the same four
    // tiles of a 24.6 MB table -- on one XCD instead of spreading each over eight caches.
    constexpr unsigned kRun = 16;
    const unsigned padded = (n_units + 8 * kRun - 1) / (8 * kRun) * (8 * kRun);
    for (unsigned v = blockIdx.x; v < padded; v += gridDim.x) {
        const unsigned xcd = v & 7u, j = v >> 3;
        const unsigned u = ((j / kRun) * 8u + xcd) * kRun + (j % kRun);
        if (u >= n_units) continue;
        const int* desc = da.chunks + 4 * (long long)(u >> 2);
        const int bucket = __builtin_amdgcn_readfirstlane(desc[0]);
        const int begin = __builtin_amdgcn_readfirstlane(desc[1]) + 16 * (int)(u & 3);
        const int cnt = __builtin_amdgcn_readfirstlane(desc[2]) - begin;       // items of this wave (may be <= 0)
        if (cnt <= 0) continue;
        const int ch = bucket >> 9, blk = bucket & 511;
        const int my_tc = da.items[begin + (slot < cnt ? slot : 0)];
        // the block's four filter tiles: issued before anything waits, consumed after the residuals are staged
        const uint4* tiles = (const uint4*)(da.detail_f32 + ((long long)ch * da.num_base + blk) * kBlockFilterTiles * 2048);
        uint4 av[kBlockFilterTiles][4];
#pragma unroll
        for (int tile = 0; tile < kBlockFilterTiles; ++tile)
#pragma unroll
            for (int q = 0; q < 4; ++q) av[tile][q] = tiles[(tile * 4 + q) * 64 + lane];
        const int rows = __builtin_amdgcn_readfirstlane(da.block_rows[blk]);
        double rv[kFilterGroup];
#pragma unroll
        for (int i = 0; i < kFilterGroup; ++i) rv[i] = da.r[(long long)__builtin_amdgcn_readlane(my_tc, i) * N + lane];
        bool nz = false;
#pragma unroll
        for (int i = 0; i < kFilterGroup; ++i) {
            s_r[i * kStageStride + lane] = rv[i];
            const bool any = __ballot(rv[i] != 0.0) != 0;
            if (slot == i) nz = any;
        }
        __syncthreads();
        bf16x8 r_hi[2], r_lo[2];
        const double ss = load_b_operand(r_hi, r_lo, s_r, lane);
        const float window = 2.0f * (kFilterSlack * (float)__builtin_sqrt(ss) + kFilterAbs);
#pragma unroll
        for (int tile = 0; tile < kBlockFilterTiles; ++tile) {
            const f32x4 acc = filter_tile_mfma(av[tile], r_hi, r_lo);
#pragma unroll
            for (int v = 0; v < 4; ++v) s_p[(tile * 16 + sub * 4 + v) * kFilterGroup + slot] = acc[v];
        }
        __syncthreads();
        float pv[16];
        float mx = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            pv[i] = fabsf(s_p[(4 * i + sub) * kFilterGroup + slot]);
            mx = fmaxf(mx, pv[i]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float thr = fmaxf(mx, da.approx_max[my_tc]) - window;      // a row far below the base maximum cannot win
        __syncthreads();                                         // s_p is dead from here: its space takes the products
        unsigned mask = 0;
        if (slot < cnt && nz) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!(pv[i] < thr)) mask |= 1u << i;
            mask &= low_bits((rows - sub + 3) >> 2);
        }
        double best_val, unused_val;
        int best_row, unused_row;
        const double* block = da.detail + ((long long)ch * da.detail_rows + __builtin_amdgcn_readfirstlane(da.block_row_off[blk])) * N;
        evaluate_survivors(s_cand, s_r, lane, mask, 0u, 0, 0, block, block, best_val, best_row, unused_val, unused_row);
        keep_better(best_val, best_row, __shfl_xor(best_val, 16), __shfl_xor(best_row, 16));
        keep_better(best_val, best_row, __shfl_xor(best_val, 32), __shfl_xor(best_row, 32));
        if (sub == 0 && slot < cnt) {
            da.cand_val[(long long)(begin + slot) * kMaxRowParts] = best_val;
            da.cand_row[(long long)(begin + slot) * kMaxRowParts] = best_row;
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
// fill: each active tile-channel drops one item per unlocked, non-repeated block (block 0 excepted) into
// that block's bucket and remembers the slot.  A slab of 256 list entries (one channel) first ranks its items
// per block in LDS, reserves one range per touched bucket with a single global atomic, then scatters: popular
// blocks would otherwise serialise tens of thousands of same-address atomics per step.
__global__ __launch_bounds__(256) void mp_fill_kernel(const Workspace ws, int cur, int cur_cnt)
{
    __shared__ unsigned s_cnt[512];
    __shared__ unsigned s_base[512];
    // Every workgroup first turns the per-(channel, block) item counts the finish kernel left into bucket ranges by an
    // exclusive scan of its own (1536 counts, six per thread: cheaper than a launch for one workgroup to do it once);
    // workgroup 0 also writes the chunk descriptors and counters the detail sweep reads, and clears the other count
    // buffer for the finish kernel that follows.
    __shared__ unsigned s_scan_items[256];
    __shared__ unsigned s_scan_chunks[256];
    __shared__ unsigned s_start[kNumBuckets];
    __shared__ unsigned s_coff[kNumBuckets + 1];
    {
        constexpr int kPer = kNumBuckets / 256;               // 6
        static_assert(kPer * 256 == kNumBuckets, "six buckets per thread");
        const int t = threadIdx.x;
        const unsigned* cnt = ws.bucket_count[cur_cnt];
        unsigned c[kPer];
        unsigned items = 0, chunks = 0;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            c[k] = cnt[t * kPer + k];
            items += c[k];
            chunks += (c[k] + kChunkItems - 1) / kChunkItems;
        }
        s_scan_items[t] = items;
        s_scan_chunks[t] = chunks;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {                    // Hillis-Steele inclusive scan
            unsigned a = 0, b = 0;
            if (t >= d) { a = s_scan_items[t - d]; b = s_scan_chunks[t - d]; }
            __syncthreads();
            s_scan_items[t] += a;
            s_scan_chunks[t] += b;
            __syncthreads();
        }
        unsigned start = s_scan_items[t] - items, coff = s_scan_chunks[t] - chunks;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            s_start[t * kPer + k] = start;
            s_coff[t * kPer + k] = coff;
            start += c[k];
            coff += (c[k] + kChunkItems - 1) / kChunkItems;
        }
        const unsigned total_chunks = s_scan_chunks[255], total_items = s_scan_items[255];
        if (t == 255) s_coff[kNumBuckets] = total_chunks;
        __syncthreads();
        if (blockIdx.x == 0) {
            unsigned* nxt = ws.bucket_count[cur_cnt ^ 1];
            for (int b = t; b < kNumBuckets; b += 256) nxt[b] = 0;
            if (t == 0) {
                ws.counters[6] = 0;                             // chunk cursor
                ws.counters[7] = total_chunks;                  // chunks this step
                ws.counters[8] = total_items;                   // items this step
            }
            for (unsigned j = (unsigned)t; j < total_chunks; j += 256) {
                int lo = 0, hi = kNumBuckets;                   // the bucket b with s_coff[b] <= j < s_coff[b + 1]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (s_coff[mid] <= j) lo = mid; else hi = mid;
                }
                const unsigned first = s_start[lo] + (j - s_coff[lo]) * kChunkItems;
                const unsigned end = (lo + 1 < kNumBuckets) ? s_start[lo + 1] : total_items;
                int* d = ws.chunks + 4 * (long long)j;
                d[0] = lo; d[1] = (int)first; d[2] = (int)(first + kChunkItems < end ? first + kChunkItems : end); d[3] = 0;
            }
        }
    }
    const int s0 = ((int)ws.counters[cur * 3 + 0] + 255) >> 8, s1 = ((int)ws.counters[cur * 3 + 1] + 255) >> 8,
              s2 = ((int)ws.counters[cur * 3 + 2] + 255) >> 8;
    for (int b = threadIdx.x; b < 512; b += 256) s_cnt[b] = 0;
    __syncthreads();
    for (int slab = blockIdx.x; slab < s0 + s1 + s2; slab += gridDim.x) {
        const int ch = slab < s0 ? 0 : (slab < s0 + s1 ? 1 : 2);
        const int n_act = (int)ws.counters[cur * 3 + ch];
        const int pos = (slab - (ch == 0 ? 0 : (ch == 1 ? s0 : s0 + s1))) * 256 + (int)threadIdx.x;
        const bool valid = pos < n_act;
        int tc = 0, nb = 0;
        if (valid) {
            tc = ws.act[cur][ch][pos];
            nb = nblk_count(ws.nblk[tc]);
            for (int i = 0; i < nb; ++i) {
                const unsigned e = ws.blk_list[(long long)tc * kMaxDeviceK + i];
                if (!entry_is_item(e)) continue;
                ws.item_slot[(long long)tc * kMaxDeviceK + i] = (int)atomicAdd(&s_cnt[entry_block(e)], 1u);   // rank within the slab
            }
        }
        __syncthreads();
        for (int b = threadIdx.x; b < 512; b += 256) {
            const unsigned c = s_cnt[b];
            if (c) {
                const int gb = (ch << 9) | b;
                s_base[b] = s_start[gb] + atomicAdd(&ws.bucket_cursor[gb], c);
                s_cnt[b] = 0;
            }
        }
        __syncthreads();
        if (valid) {
            for (int i = 0; i < nb; ++i) {
                const unsigned e = ws.blk_list[(long long)tc * kMaxDeviceK + i];
                if (!entry_is_item(e)) continue;
                const unsigned slot = s_base[entry_block(e)] + (unsigned)ws.item_slot[(long long)tc * kMaxDeviceK + i];
                ws.items[slot] = tc;
                ws.item_slot[(long long)tc * kMaxDeviceK + i] = (int)slot;
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------------
// finish: per active tile-channel -- argmax over the whole dynamic dictionary in index order (strict '>',
// first maximum wins: MatchingPursuit.cpp:14-19), delta/zigzag, quantise, record, residual update
// (mathvector.cpp:116-148), unlock DetailBasis[choice] (CompressedImage.cpp:226-229), termination.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mp_finish_kernel(const Workspace ws, const DictDevice dict, const Outputs out,
                                                       const double* __restrict__ quant, int K, int step, int cur,
                                                       int max_parts, int max_row_parts, int grid_hint)
{
    const int parts = effective_parts((((int)ws.counters[cur * 3 + 0] + 63) >> 6) + (((int)ws.counters[cur * 3 + 1] + 63) >> 6) +
                                          (((int)ws.counters[cur * 3 + 2] + 63) >> 6), grid_hint, max_parts);
    const int row_parts = effective_parts((int)ws.counters[7], grid_hint, max_row_parts);
    const int s0 = ((int)ws.counters[cur * 3 + 0] + 255) >> 8, s1 = ((int)ws.counters[cur * 3 + 1] + 255) >> 8,
              s2 = ((int)ws.counters[cur * 3 + 2] + 255) >> 8;
    __shared__ unsigned s_cnt[512];          // next step's items per block, counted per slab before going global
    for (int b = threadIdx.x; b < 512; b += 256) s_cnt[b] = 0;
    if (blockIdx.x == 0)                     // the fill kernel of this step is done with its bucket cursors: clear for the next
        for (int b = threadIdx.x; b < kNumBuckets; b += 256) ws.bucket_cursor[b] = 0;
    __syncthreads();
    for (int slab = blockIdx.x; slab < s0 + s1 + s2; slab += gridDim.x) {
    const int ch = slab < s0 ? 0 : (slab < s0 + s1 ? 1 : 2);
    const int n_act = (int)ws.counters[cur * 3 + ch];
    const int pos = (slab - (ch == 0 ? 0 : (ch == 1 ? s0 : s0 + s1))) * 256 + (int)threadIdx.x;
    if (pos < n_act) {
    const int tc = ws.act[cur][ch][pos];
    const unsigned oi = (unsigned)ws.out_index[tc];
    const long long rec = (long long)(oi & 0x3FFFFFFFu);
    const int nblk_word = ws.nblk[tc];
    const int nb = nblk_count(nblk_word);
    const int extra = ws.extra_rows[tc];

    double best_val = 0.0;
    int best_idx = -1;
    int best_sel = 0;                       // ~idx for a base atom, row index into `detail` otherwise
    for (int p = 0; p < parts; ++p) {
        const double v = ws.part_val[(long long)tc * kMaxParts + p];
        const int i = ws.part_idx[(long long)tc * kMaxParts + p];
        if (i >= 0 && __builtin_fabs(v) > __builtin_fabs(best_val)) { best_val = v; best_idx = i; best_sel = ~i; }
    }
    int off = dict.num_base;
    for (int i = 0; i < nb; ++i) {
        const unsigned e = ws.blk_list[(long long)tc * kMaxDeviceK + i];
        const int blk = entry_block(e);
        const int rows = entry_rows(e);
        if (!entry_repeat(e)) {             // a repeated block can never win: identical projections at a higher index
            const double* cv;
            const int* cr;
            int nparts = row_parts;
            if (blk == 0) { cv = ws.cand0_val + tc; cr = ws.cand0_row + tc; nparts = 1; }
            else {
                const long long slot = ws.item_slot[(long long)tc * kMaxDeviceK + i];
                cv = ws.cand_val + slot * kMaxRowParts;
                cr = ws.cand_row + slot * kMaxRowParts;
            }
            for (int p = 0; p < nparts; ++p) {          // row ranges in ascending order
                const double v = cv[p];
                const int row = cr[p];
                if (row >= 0 && __builtin_fabs(v) > __builtin_fabs(best_val)) {
                    best_val = v; best_idx = off + row; best_sel = dict.block_row_off[blk] + row;
                }
            }
        }
        off += rows;
    }
    const unsigned swept = ws.swept[tc] + (unsigned)(dict.num_base + extra);
    ws.swept[tc] = swept;

    bool done = false;
    int count = 0;
    unsigned record = 0;
    bool unlocked = false;
    double coeff = 0.0;
    if (best_idx < 0) {                                       // MatchingPursuit.cpp:50-54
        done = true; count = step;
    } else {
        const int prev = ws.prev_id[tc];
        const unsigned delta = (step > 0) ? (((unsigned)(best_idx - prev) << 1) ^ (unsigned)((best_idx - prev) >> 31))
                                          : (unsigned)best_idx;
        ws.prev_id[tc] = best_idx;
        const double qstep = quant[ch * K + step];
        const int q = (int)__builtin_round(best_val / qstep);
        const unsigned zz = ((unsigned)q << 1) ^ (unsigned)(q >> 31);
        record = (delta & 0xFFFFu) | ((zz & 0xFFFFu) << 16);
        if (q == 0) {                                         // :66-69
            done = true; count = step;
        } else {
            coeff = qstep * (double)q;                        // residual update itself: mp_update_kernel (coalesced)
            if (best_idx < dict.num_base) {                   // unlock DetailBasis[choice]; duplicates are appended again
                const int new_rows = dict.block_rows[best_idx];
                unsigned e = (unsigned)best_idx | ((unsigned)new_rows << 9);
                for (int i = 0; i < nb; ++i)
                    if (entry_block(ws.blk_list[(long long)tc * kMaxDeviceK + i]) == best_idx) e |= 0x8000u;
                ws.blk_list[(long long)tc * kMaxDeviceK + nb] = (uint16_t)e;
                ws.nblk[tc] = (nb + 1) | (nblk_word & 0x100) | (best_idx == 0 ? 0x100 : 0);
                ws.extra_rows[tc] = extra + new_rows;
                unlocked = true;
            }
            if (step + 1 == K) { done = true; count = K; }
        }
    }
    out.choices[rec * K + step] = record;
    ws.upd_coeff[tc] = coeff;
    ws.upd_sel[tc] = best_sel;
    if (done) {
        // the energy of the final residual (diagnostic output) is summed by mp_energy_kernel after the last step: the
        // residual row of a finished tile-channel does not change any more and its last update stays in upd_coeff/upd_sel
        out.counts[rec] = (uint16_t)count;
        if (out.swept) out.swept[rec] = swept;
    } else {
        // wave-aggregated append to the channel's next active list (the wave is channel-pure)
        const unsigned long long going = __ballot(true);
        const int leader = __builtin_ctzll(going);
        unsigned first = 0;
        if ((int)(threadIdx.x & 63) == leader) first = atomicAdd(&ws.counters[(cur ^ 1) * 3 + ch], (unsigned)__popcll(going));
        first = (unsigned)__builtin_amdgcn_readlane((int)first, leader);
        const unsigned slot = first + (unsigned)__popcll(going & ((1ULL << (threadIdx.x & 63)) - 1ULL));
        ws.act[cur ^ 1][ch][slot] = tc;
        // items of the next step: every unlocked, non-repeated block except block 0
        const int nb2 = nb + (unlocked ? 1 : 0);
        for (int i = 0; i < nb2; ++i) {
            const unsigned e = ws.blk_list[(long long)tc * kMaxDeviceK + i];
            if (!entry_is_item(e)) continue;
            atomicAdd(&s_cnt[entry_block(e)], 1u);
        }
    }
    }
    __syncthreads();
    {
        unsigned* cnt = ws.bucket_count[(step + 1) & 1];
        for (int b = threadIdx.x; b < 512; b += 256) {
            const unsigned c = s_cnt[b];
            if (c) { atomicAdd(&cnt[(ch << 9) | b], c); s_cnt[b] = 0; }
        }
    }
    __syncthreads();
    }
}


// update: r -= coeff * atom for every tile-channel the finish kernel quantised to a non-zero coefficient
// (Vector::Scale then Vector::Subtract, mathvector.cpp:116-148: two roundings).  One wave per 64 entries of
// the active list, LANE = PIXEL: each residual and atom row is one coalesced 512-byte access.
__global__ __launch_bounds__(64) void mp_update_kernel(const Workspace ws, const DictDevice dict, int cur)
{
    const int lane = threadIdx.x;
    const UnitMap m = unit_map(ws.counters, cur);
    const int total_groups = m.groups();
    for (int gi = blockIdx.x; gi < total_groups; gi += gridDim.x) {
    int ch, group;
    decode_group(m, gi, ch, group);
    const int n_act = (int)scalar_counter(ws.counters, cur * 3 + ch);
    const int pos = group * 64 + lane;
    int tc = 0, sel = 0;
    double coeff = 0.0;
    if (pos < n_act) {
        tc = ws.act[cur][ch][pos];
        coeff = ws.upd_coeff[tc];
        sel = ws.upd_sel[tc];
    }
    if (!__ballot(coeff != 0.0)) continue;
    const int c_lo = __double2loint(coeff), c_hi = __double2hiint(coeff);
    // fixed batches of 8 tile-channels: 16 independent coalesced loads in flight, then the stores
    // (entries that do not update keep coeff 0 and are not stored)
#pragma unroll 1
    for (int base = 0; base < 64; base += 8) {
        double rv[8], av[8], cv[8];
        double* rp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int p = base + k;
            const int tci = __builtin_amdgcn_readlane(tc, p);
            const int seli = __builtin_amdgcn_readlane(sel, p);
            cv[k] = __hiloint2double(__builtin_amdgcn_readlane(c_hi, p), __builtin_amdgcn_readlane(c_lo, p));
            const double* row = (seli < 0) ? (dict.base + (long long)(~seli) * N)
                                           : (dict.detail + ((long long)ch * dict.detail_rows + seli) * N);
            rp[k] = ws.r + (long long)tci * N + lane;
            rv[k] = *rp[k];
            av[k] = row[lane];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (cv[k] != 0.0) {
                const double scaled = cv[k] * av[k];      // Vector::Scale
                *rp[k] = rv[k] - scaled;                   // Vector::Subtract
            }
        }
    }
    }
}

// Per-stream symbol histograms for the Huffman/Golomb stage (RCCL all-reduce input):
// stream 0 = lengths, stream 1 + 2K*ch + 2i (+1) = codes[2K*ch + 2i (+1)] (CompressedImage.cpp:556-572).
// Block (job, slice): job 0 counts the lengths of all tile-channels, job 1 + ch*K + i counts the deltaId and
// intCoeff symbols of step i of channel ch; a job's tiles are split over `slices` blocks.  Counting is done in
// LDS (two 8192-bin tables), only non-zero bins are flushed with global atomics.
__global__ __launch_bounds__(256) void mp_histogram_kernel(const HistParams p, int slices)
{
    __shared__ unsigned h0[kHistBins];
    __shared__ unsigned h1[kHistBins];
    const int job = blockIdx.x / slices;
    const int slice = blockIdx.x - job * slices;
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) { h0[b] = 0; h1[b] = 0; }
    __syncthreads();
    const long long per = (p.tiles + slices - 1) / slices;
    const long long t0 = slice * per;
    const long long t1 = (t0 + per < p.tiles) ? t0 + per : p.tiles;
    if (job == 0) {
        for (long long o = t0 * 3 + threadIdx.x; o < t1 * 3; o += blockDim.x)
            if (p.counts[o] < kHistBins) atomicAdd(&h0[p.counts[o]], 1u);
    } else {
        const int ch = (job - 1) / p.K, i = (job - 1) - ch * p.K;
        for (long long t = t0 + threadIdx.x; t < t1; t += blockDim.x) {
            const long long o = t * 3 + ch;
            if (p.counts[o] > i) {
                const uint32_t rec = p.choices[o * p.K + i];
                if ((rec & 0xFFFFu) < (unsigned)kHistBins) atomicAdd(&h0[rec & 0xFFFFu], 1u);       // symbols beyond the table are
                if ((rec >> 16) < (unsigned)kHistBins) atomicAdd(&h1[rec >> 16], 1u);               // not counted (sharding.py agrees)
            }
        }
    }
    __syncthreads();
    uint32_t* g0;
    uint32_t* g1 = nullptr;
    if (job == 0) g0 = p.hist;
    else {
        const int ch = (job - 1) / p.K, i = (job - 1) - ch * p.K;
        g0 = p.hist + (long long)(1 + 2 * p.K * ch + 2 * i) * kHistBins;
        g1 = g0 + kHistBins;
    }
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) {
        if (h0[b]) atomicAdd(&g0[b], h0[b]);
        if (g1 && h1[b]) atomicAdd(&g1[b], h1[b]);
    }
}

// --------------------------------------------------------------------------------------------------
// decode: one wave per tile, LANE = PIXEL.  For each channel the recorded steps are replayed in order:
// choice = running sum of the zig-zag deltas, coefficient = quant[i] * zigzagDecode(intCoeff), and every pixel
// accumulates results[j] += basis[j] * coeff in step order (MatchingPursuit.cpp:109-128, one rounding for the
// product, one for the sum -- no FMA).  The dynamic dictionary is resolved through the same block list as in
// the encoder (base rows, then DetailBasis[choice] of every earlier base choice, repeats included,
// CompressedImage.cpp:219-248).  Then RGBFromYUV (misc.cpp:28-36): round half away, clamp, store.
// --------------------------------------------------------------------------------------------------
// T = float: FromCoeffsDynamicFast (MatchingPursuit.cpp:130-147) on the dictionary rounded to float; RGBFromYUV on the floats
// widened to double (CompressedImage.cpp:877-881).
template <class T>
__global__ __launch_bounds__(64) void mp_decode_kernel(const DictDevice dict, const DecodeParams p)
{
    constexpr bool kFast = std::is_same<T, float>::value;
    const T* const base_rows = reinterpret_cast<const T*>(kFast ? static_cast<const void*>(dict.base32) : static_cast<const void*>(dict.base));
    const T* const detail_rows = reinterpret_cast<const T*>(kFast ? static_cast<const void*>(dict.detail32) : static_cast<const void*>(dict.detail));
    const int lane = threadIdx.x;
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
        double yuv[3];
        for (int ch = 0; ch < 3; ++ch) {
            const int count = p.counts[t * 3 + ch];
            const uint32_t* rec = p.choices + (t * 3 + ch) * p.K;
            T acc = 0;
            int choice = 0;
            for (int i = 0; i < count && i < p.K; ++i) {
                const uint32_t r = rec[i];
                const unsigned delta = r & 0xFFFFu, zz = r >> 16;
                const int d = (int)((delta >> 1) ^ (0u - (delta & 1u)));
                choice = (i > 0) ? choice + d : (int)delta;
                const int q = (int)((zz >> 1) ^ (0u - (zz & 1u)));
                const T coeff = (T)p.quant[ch * p.K + i] * (T)q;
                // locate row `choice` in the dynamic dictionary of this tile-channel; the reference builds it from
                // ALL `count` choices before summing (FromCoeffsDynamic :111), so resolve against the full list
                const T* row = nullptr;
                if (choice >= 0 && choice < dict.num_base) {
                    row = base_rows + (long long)choice * N;
                } else if (choice >= dict.num_base) {
                    int off = dict.num_base;
                    int walk = 0;
                    for (int k = 0; k < count && k < p.K; ++k) {
                        const uint32_t rk = rec[k];
                        const unsigned dk = rk & 0xFFFFu;
                        walk = (k > 0) ? walk + (int)((dk >> 1) ^ (0u - (dk & 1u))) : (int)dk;
                        if (walk >= 0 && walk < dict.num_base) {
                            const int rows = dict.block_rows[walk];
                            if (choice < off + rows) {
                                row = detail_rows + ((long long)ch * dict.detail_rows + dict.block_row_off[walk] + (choice - off)) * N;
                                break;
                            }
                            off += rows;
                        }
                    }
                }
                if (!row) {
                    if (lane == 0) *p.error_flag = 1;
                    continue;
                }
                const T term = row[lane] * coeff;
                acc = acc + term;
            }
            yuv[ch] = (double)acc;
        }
        const int tx = (int)(t / p.tiles_y), ty = (int)(t - (long long)tx * p.tiles_y);
        const int u = tx * 8 + (lane & 7), v = ty * 8 + (lane >> 3);          // block index = dx + 8*dy
        if (u < p.width && v < p.height) {
            const double Y = yuv[0], U = yuv[1], V = yuv[2];
            double rr = __builtin_round(Y + 1.13983 * V);
            double gg = __builtin_round(Y - 0.39466 * U - 0.58060 * V);
            double bb = __builtin_round(Y + 2.03211 * U);
            rr = rr < 0.0 ? 0.0 : (rr > 255.0 ? 255.0 : rr);
            gg = gg < 0.0 ? 0.0 : (gg > 255.0 ? 255.0 : gg);
            bb = bb < 0.0 ? 0.0 : (bb > 255.0 ? 255.0 : bb);
            uint8_t* px = p.rgb + 3 * ((long long)v * p.width + u);
            px[0] = (uint8_t)rr;
            px[1] = (uint8_t)gg;
            px[2] = (uint8_t)bb;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// host side: workspace carving and the per-step launch sequence
// --------------------------------------------------------------------------------------------------
namespace {
size_t align_up(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

struct Carver {
    char* base;
    size_t off = 0;
    template <class T>
    T* take(size_t count) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += align_up(count * sizeof(T));
        return p;
    }
};

Workspace carve(char* mem, int cap, int K, size_t* total)
{
    Carver c{mem};
    Workspace w{};
    const size_t n = static_cast<size_t>(cap);
    w.cap = cap;
    w.max_items = static_cast<long long>(n) * (K > 1 ? K - 1 : 1);
    w.max_chunks = static_cast<int>(w.max_items / kChunkItems) + kNumBuckets + 1;
    w.r = c.take<double>(n * N);
    w.part_val = c.take<double>(n * kMaxParts);
    w.part_idx = c.take<int>(n * kMaxParts);
    w.prev_id = c.take<int>(n);
    w.nblk = c.take<int>(n);
    w.extra_rows = c.take<int>(n);
    w.swept = c.take<unsigned>(n);
    w.blk_list = c.take<uint16_t>(n * kMaxDeviceK);
    w.item_slot = c.take<int>(n * kMaxDeviceK);
    w.out_index = c.take<int>(n);
    for (int a = 0; a < 2; ++a)
        for (int ch = 0; ch < 3; ++ch) w.act[a][ch] = c.take<int>(n);
    w.cand0_val = c.take<double>(n);
    w.cand0_row = c.take<int>(n);
    w.approx_max = c.take<float>(n);
    w.upd_coeff = c.take<double>(n);
    w.upd_sel = c.take<int>(n);
    w.counters = c.take<unsigned>(16);
    w.bucket_count[0] = c.take<unsigned>(kNumBuckets);
    w.bucket_count[1] = c.take<unsigned>(kNumBuckets);
    w.bucket_cursor = c.take<unsigned>(kNumBuckets);
    w.chunks = c.take<int>(static_cast<size_t>(w.max_chunks) * 4);
    w.items = c.take<int>(static_cast<size_t>(w.max_items));
    w.cand_val = c.take<double>(static_cast<size_t>(w.max_items) * kMaxRowParts);
    w.cand_row = c.take<int>(static_cast<size_t>(w.max_items) * kMaxRowParts);
    if (total) *total = c.off;
    return w;
}
}  // namespace

size_t workspace_bytes(int cap, int K)
{
    size_t total = 0;
    carve(nullptr, cap, K, &total);
    return total;
}

Workspace carve_workspace(void* device_mem, int cap, int K) { return carve(static_cast<char*>(device_mem), cap, K, nullptr); }

namespace {
FilterArgs filter_args(const Workspace& ws, const DictDevice& dict, int cur)
{
    FilterArgs fa{};
    fa.counters_in = ws.counters;
    fa.counters_out = ws.counters;
    for (int ch = 0; ch < 3; ++ch) {
        fa.act[ch] = ws.act[cur][ch];
        fa.block0[ch] = dict.detail + (long long)ch * dict.detail_rows * N;
        fa.block0_f32[ch] = dict.detail_f32 + (long long)ch * dict.num_base * kBlockFilterTiles * 2048;
    }
    fa.r = ws.r;
    fa.upd_coeff = ws.upd_coeff;
    fa.upd_sel = ws.upd_sel;
    fa.nblk = ws.nblk;
    fa.part_val = ws.part_val;
    fa.part_idx = ws.part_idx;
    fa.cand0_val = ws.cand0_val;
    fa.cand0_row = ws.cand0_row;
    fa.approx_max = ws.approx_max;
    fa.base = dict.base;
    fa.base_f32 = dict.base_f32;
    fa.num_base = dict.num_base;
    fa.rows0 = dict.block0_rows;
    return fa;
}
}  // namespace

// Final residual energies (Outputs::energy, a diagnostic the reference does not compute), once per batch after the last
// step.  A wave takes 16 tile-channels: lane = pixel for coalesced reads of the residual rows, eight rows in flight (and
// the atom of a still pending update, count == K, folded in with the update's two roundings); the squares go through LDS
// so that lane k adds tile-channel k's 64 terms in the oracle's order.
constexpr int kEnergyGroup = 16;
__global__ __launch_bounds__(64) void mp_energy_kernel(const Workspace ws, const DictDevice dict, const Outputs out, int n)
{
    __shared__ double sq[kEnergyGroup * 65];
    const int lane = (int)threadIdx.x;
    for (long long first = (long long)blockIdx.x * kEnergyGroup; first < n; first += (long long)gridDim.x * kEnergyGroup) {
        const int here = (int)(n - first < kEnergyGroup ? n - first : kEnergyGroup);
        const long long mine = first + (lane < here ? lane : 0);
        const double my_coeff = ws.upd_coeff[mine];
        const int my_sel = ws.upd_sel[mine];
        const unsigned my_out = (unsigned)ws.out_index[mine];
        for (int i0 = 0; i0 < here; i0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < here ? i0 + u : here - 1;
                v[u] = ws.r[(first + i) * N + lane];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < here ? i0 + u : here - 1;
                const double coeff = __shfl(my_coeff, i);
                if (coeff != 0.0) {
                    const int sel = __shfl(my_sel, i);
                    const int ch = (int)((unsigned)__shfl((int)my_out, i) >> 30);
                    const double* row = (sel < 0) ? (dict.base + (long long)(~sel) * N)
                                                  : (dict.detail + ((long long)ch * dict.detail_rows + sel) * N);
                    const double scaled = coeff * row[lane];
                    v[u] = v[u] - scaled;
                }
                sq[i * 65 + lane] = v[u] * v[u];
            }
        }
        __syncthreads();
        if (lane < here) {
            double e2 = 0.0;
            for (int j = 0; j < N; ++j) e2 += sq[lane * 65 + j];
            out.energy[(long long)(my_out & 0x3FFFFFFFu)] = e2;
        }
        __syncthreads();
    }
}

int enqueue_pursuit(const DictDevice& dict, const Workspace& ws, const FrameInput& in, const Outputs& out,
                    const double* quant_dev, int K, long long tc_begin, int n, int parts, int row_parts, int sweep_waves,
                    void* stream_, void** base_events, void* side_stream_, void* fork_event_, void* join_event_)
{
    DetailFilterArgs detail_args{};
    detail_args.counters = ws.counters;
    detail_args.chunks = ws.chunks;
    detail_args.items = ws.items;
    detail_args.r = ws.r;
    detail_args.cand_val = ws.cand_val;
    detail_args.cand_row = ws.cand_row;
    detail_args.approx_max = ws.approx_max;
    detail_args.detail = dict.detail;
    detail_args.detail_rows = dict.detail_rows;
    detail_args.detail_f32 = dict.detail_f32;
    detail_args.block_rows = dict.block_rows;
    detail_args.block_row_off = dict.block_row_off;
    detail_args.num_base = dict.num_base;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipStream_t side = static_cast<hipStream_t>(side_stream_);
    hipEvent_t ev_fork = static_cast<hipEvent_t>(fork_event_), ev_join = static_cast<hipEvent_t>(join_event_);
    const bool forked = side != nullptr && ev_fork != nullptr && ev_join != nullptr;
    if (n < 1 || n > ws.cap) return (int)hipErrorInvalidValue;
    const bool filtered = parts < 1;          // sweeps through the MFMA filter instead of correlating every row exactly
    if (filtered) row_parts = 1;
    if (parts < 1) parts = 1;
    if (parts > kMaxParts) parts = kMaxParts;
    if (row_parts < 1) row_parts = 1;
    if (row_parts > kMaxRowParts) row_parts = kMaxRowParts;
    // Persistent grids: every kernel strides over work whose amount is only known on the device (active
    // counts shrink every step), so grid sizes are machine-sized, not problem-sized.
    const int per_list = in.vec_in ? n : (n + 2) / 3;
    const unsigned lists = in.vec_in ? 1u : 3u;
    const unsigned max_groups = lists * (unsigned)((per_list + 63) / 64);
    const unsigned max_slabs = lists * (unsigned)((per_list + 255) / 256);
    const unsigned slots = (unsigned)(sweep_waves > 0 ? sweep_waves : 3072);
    auto clampu = [](unsigned v, unsigned hi) { return v < hi ? (v ? v : 1u) : hi; };
    // the wave-autonomous filter kernels hold 2 waves per SIMD; experiments: MPC_WAVE_SLOTS, MPC_GROUPS
    static const unsigned wave_slots = [] { const char* v = std::getenv("MPC_WAVE_SLOTS"); return v && *v ? (unsigned)std::atoi(v) : 2048u; }();
    // at most two column groups per wave would be used anyway: take the variant that spends its registers on tile reads
    const bool small_batch = (long long)n <= 2LL * 16 * wave_slots && std::getenv("MPC_NO_SMALL") == nullptr;
    static const int force_groups = [] { const char* v = std::getenv("MPC_GROUPS"); return v && *v ? std::atoi(v) : 0; }();
    hipLaunchKernelGGL(mp_init_kernel, dim3(clampu((unsigned)((n + 2) / 3), 16384u)), dim3(64), 0, s, ws, in, tc_begin, n);
    for (int step = 0; step < K; ++step) {
        const int cur = step & 1;
        if (step > 0) {
            hipLaunchKernelGGL(mp_fill_kernel, dim3(clampu(max_slabs, 2048u)), dim3(256), 0, s, ws, cur, step & 1);
            if (forked && !filtered) (void)hipStreamWaitEvent(s, ev_join, 0);   // the previous step's residual update
        }
        if (base_events) (void)hipEventRecord(static_cast<hipEvent_t>(base_events[2 * step]), s);
        if (filtered)
            if (small_batch)
                hipLaunchKernelGGL((mp_filter_wave_kernel<2, 4>), dim3(clampu(lists * (unsigned)((per_list + 15) / 16), wave_slots)), dim3(64),
                                   0, s, filter_args(ws, dict, cur), cur, step > 0 ? 1 : 0, force_groups);
            else
                hipLaunchKernelGGL((mp_filter_wave_kernel<4, 1>), dim3(clampu(lists * (unsigned)((per_list + 15) / 16), wave_slots)), dim3(64),
                                   0, s, filter_args(ws, dict, cur), cur, step > 0 ? 1 : 0, force_groups);
        else
            hipLaunchKernelGGL(mp_base_kernel, dim3(clampu(max_groups * (unsigned)(parts + 1), slots)), dim3(64), 0, s, ws, dict,
                               cur, parts, step > 0 ? 1 : 0, (int)slots);
        if (base_events) (void)hipEventRecord(static_cast<hipEvent_t>(base_events[2 * step + 1]), s);
        if (step > 0) {
            if (filtered)
                hipLaunchKernelGGL(mp_detail_filter_kernel, dim3(clampu(max_groups * 4u, 3072u)), dim3(64), 0, s, detail_args);
            else
                hipLaunchKernelGGL(mp_detail_kernel, dim3(clampu(max_groups * (unsigned)row_parts, slots)), dim3(64), 0, s, ws,
                                   dict, row_parts, (int)slots, ws.cand_val);
        }
        hipLaunchKernelGGL(mp_finish_kernel, dim3(clampu(max_slabs, 2048u)), dim3(256), 0, s, ws, dict, out, quant_dev, K,
                           step, cur, parts, row_parts, (int)slots);
        if (step + 1 < K) {
            // The residual update and the next step's bucket + fill only depend on the finish kernel, not on each other:
            // with a side stream the update runs beside them and is joined in front of the next sweep.
            if (forked && !filtered) {
                (void)hipEventRecord(ev_fork, s);
                (void)hipStreamWaitEvent(side, ev_fork, 0);
                hipLaunchKernelGGL(mp_update_kernel, dim3(clampu(max_groups, 4096u)), dim3(64), 0, side, ws, dict, cur);
                (void)hipEventRecord(ev_join, side);
            } else if (!filtered) {                               // the filter kernel applies the update as it reads the residuals
                hipLaunchKernelGGL(mp_update_kernel, dim3(clampu(max_groups, 4096u)), dim3(64), 0, s, ws, dict, cur);
            }
        }
    }
    if (out.energy)
        hipLaunchKernelGGL(mp_energy_kernel, dim3(clampu((unsigned)((n + kEnergyGroup - 1) / kEnergyGroup), 8192u)), dim3(64), 0, s, ws, dict, out, n);
    return (int)hipGetLastError();
}

// ---- records [tiles][3][K] -> planar [3][K][tiles] ----
// For the host entropy stage of mpc_encode_image(s): every (channel, step) stream is then one contiguous run of the
// download instead of one word per 12*K bytes.  64 tiles per block through LDS, both sides coalesced.
__global__ void __launch_bounds__(256) mp_planar_records_kernel(const uint32_t* __restrict__ choices, uint32_t* __restrict__ planar,
                                                                int tiles, int K) {
    extern __shared__ uint32_t planar_lds[];            // [64][3K + 1]
    const int rows = 3 * K, pitch = rows + 1;
    const long long t0 = static_cast<long long>(blockIdx.x) * 64;
    const int nt = static_cast<int>(min(64LL, static_cast<long long>(tiles) - t0));
    const uint32_t* src = choices + t0 * rows;
    for (int idx = threadIdx.x; idx < nt * rows; idx += 256) {
        const int tt = idx / rows, r = idx - tt * rows;
        planar_lds[tt * pitch + r] = src[idx];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < rows * 64; idx += 256) {
        const int r = idx >> 6, tt = idx & 63;
        if (tt < nt) planar[static_cast<size_t>(r) * tiles + t0 + tt] = planar_lds[tt * pitch + r];
    }
}

int launch_planar_records(const uint32_t* choices, uint32_t* planar, long long tiles, int K, void* stream) {
    if (tiles <= 0) return 0;
    const unsigned blocks = static_cast<unsigned>((tiles + 63) / 64);
    const size_t lds = sizeof(uint32_t) * 64 * (3 * static_cast<size_t>(K) + 1);
    hipLaunchKernelGGL(mp_planar_records_kernel, dim3(blocks), dim3(256), lds, static_cast<hipStream_t>(stream), choices, planar,
                       static_cast<int>(tiles), K);
    return static_cast<int>(hipGetLastError());
}

int launch_decode(const DictDevice& dict, const DecodeParams& p, void* stream)
{
    const long long tiles = (long long)p.tiles_x * p.tiles_y;
    unsigned blocks = (unsigned)(tiles < 16384 ? tiles : 16384);
    if (blocks < 1) blocks = 1;
    if (p.fast) hipLaunchKernelGGL(mp_decode_kernel<float>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, dict, p);
    else hipLaunchKernelGGL(mp_decode_kernel<double>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, dict, p);
    return (int)hipGetLastError();
}

int launch_histogram(const HistParams& p, void* stream)
{
    const int jobs = 1 + 3 * p.K;
    int slices = (int)((p.tiles + 8191) / 8192);       // ~8k tiles per block
    if (slices < 1) slices = 1;
    if (slices > 64) slices = 64;
    hipLaunchKernelGGL(mp_histogram_kernel, dim3((unsigned)(jobs * slices)), dim3(256), 0, (hipStream_t)stream, p, slices);
    return (int)hipGetLastError();
}

}  // namespace mpc
