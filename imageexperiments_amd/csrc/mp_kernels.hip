// mp_kernels.hip -- product: the decoder, the symbol histogram, the planar re-ordering of records, and the product's own
// CROSS-CHECK of the tile encoder: a step-synchronous pursuit that correlates EVERY dictionary row in the reference's arithmetic
// (MPC_PATH=steps / MPC_FILTER=0; tests/test_gpu_parity.py compares it with the persistent kernel of mp_pursuit.hip, which
// screens the rows first).  Integer outputs must equal the reference's double path (MatchingPursuit.cpp:39-74) bit for bit, so
// every value that decides anything is IEEE double with separately rounded mul/add (-ffp-contract=off) in the reference's order:
//     tot = 0; tot += row[j] * r[j]   (j ascending, mathmatrix.cpp:436-444)
// Exhaustive sweeps: r (one tile-channel per lane) lives in 128 VGPRs, the row is wave-uniform and arrives through the scalar
// data cache (s_load_dwordx16 -> SGPR operand of v_mul_f64).  SMEM returns out of order, so the only usable wait is
// lgkmcnt(0): the loop keeps exactly ONE 16-double group load in flight (mp_base_kernel, mp_detail_kernel; measured:
// tools/ubench_scalar_sweep.hip).  Per MP step: init / fill (bucket the active tile-channels by unlocked block), base sweep,
// detail sweep, finish (argmax in dictionary order, quantise, record, unlock), update.  (The step-synchronous MFMA filter
// kernels of round 1 are gone: the persistent kernel replaced them.)
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
    hipStream_t s = static_cast<hipStream_t>(stream_);
    hipStream_t side = static_cast<hipStream_t>(side_stream_);
    hipEvent_t ev_fork = static_cast<hipEvent_t>(fork_event_), ev_join = static_cast<hipEvent_t>(join_event_);
    const bool forked = side != nullptr && ev_fork != nullptr && ev_join != nullptr;
    if (n < 1 || n > ws.cap) return (int)hipErrorInvalidValue;
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
    hipLaunchKernelGGL(mp_init_kernel, dim3(clampu((unsigned)((n + 2) / 3), 16384u)), dim3(64), 0, s, ws, in, tc_begin, n);
    for (int step = 0; step < K; ++step) {
        const int cur = step & 1;
        if (step > 0) {
            hipLaunchKernelGGL(mp_fill_kernel, dim3(clampu(max_slabs, 2048u)), dim3(256), 0, s, ws, cur, step & 1);
            if (forked) (void)hipStreamWaitEvent(s, ev_join, 0);   // the previous step's residual update
        }
        if (base_events) (void)hipEventRecord(static_cast<hipEvent_t>(base_events[2 * step]), s);
        hipLaunchKernelGGL(mp_base_kernel, dim3(clampu(max_groups * (unsigned)(parts + 1), slots)), dim3(64), 0, s, ws, dict,
                           cur, parts, step > 0 ? 1 : 0, (int)slots);
        if (base_events) (void)hipEventRecord(static_cast<hipEvent_t>(base_events[2 * step + 1]), s);
        if (step > 0)
            hipLaunchKernelGGL(mp_detail_kernel, dim3(clampu(max_groups * (unsigned)row_parts, slots)), dim3(64), 0, s, ws,
                               dict, row_parts, (int)slots, ws.cand_val);
        hipLaunchKernelGGL(mp_finish_kernel, dim3(clampu(max_slabs, 2048u)), dim3(256), 0, s, ws, dict, out, quant_dev, K,
                           step, cur, parts, row_parts, (int)slots);
        if (step + 1 < K) {
            // The residual update and the next step's bucket + fill only depend on the finish kernel, not on each other:
            // with a side stream the update runs beside them and is joined in front of the next sweep.
            if (forked) {
                (void)hipEventRecord(ev_fork, s);
                (void)hipStreamWaitEvent(side, ev_fork, 0);
                hipLaunchKernelGGL(mp_update_kernel, dim3(clampu(max_groups, 4096u)), dim3(64), 0, side, ws, dict, cur);
                (void)hipEventRecord(ev_join, side);
            } else {
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
