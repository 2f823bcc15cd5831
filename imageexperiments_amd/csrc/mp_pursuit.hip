// mp_pursuit.hip -- product: the PERSISTENT matching-pursuit kernel of the MI355X (gfx950, wave64) tile encoder.
//
// One launch runs the whole K-step pursuit (CalcMPDynamic, MatchingPursuit.cpp:39-74) of every tile-channel of a batch (its
// workgroups are split over the three channels); nothing of a tile-channel's state leaves the chip between steps:
//   * a workgroup (12 waves, one per CU) keeps the filter copy of the 510 base rows and of DetailBasis[0] of its channel in
//     LDS (36 tiles of 16 rows x 64 pixels, split-bfloat16 in MFMA operand order, 144 KiB) for its whole life;
//   * a wave owns kGroups (= 1) column groups of 16 tile-channels; lane (slot = l & 15, h = l >> 4) holds pixels 16h .. 16h+15 of the
//     residual of tile-channel `slot` in double (32 VGPRs per group) -- which is at once the B operand slice of
//     v_mfma_f32_16x16x32_bf16 (k order 16h + 8kk + j, host_dictionary.h: filter_tiles k_order 1) and a quarter of the
//     reference's sequential dot product;
//   * per step: split-bf16 approximations of all 510 (+63) rows from LDS on the matrix cores with top-two tracking; the few
//     rows that can be the maximum ("survivors", same proven window as mp_kernels.hip: filtered sweeps) are evaluated exactly:
//     the four lanes of a slot each multiply their 16 pixels and pass the running sum on in pixel order, so the sum is the
//     reference's `tot += l*r` chain (mathmatrix.cpp:436-444), double, no FMA, j ascending;
//   * argmax in dictionary order, delta / zig-zag, quantise, record, residual update (two roundings, mathvector.cpp:116-148)
//     in registers; a finished group of 16 is replaced from a device-side queue, so groups of one wave are at different steps.
// Detail blocks other than DetailBasis[0] ("pairs": a tile-channel and a block it has unlocked) differ per tile-channel, so
// their rows cannot share MFMA operands.  Their approximations are instead KEPT (64 floats per pair, L2-resident scratch)
// and updated with the Gram matrix: after choosing atom a with coefficient c the projection of every row b changes by exactly
// -c <b, a>, so P_b -= float(c) * G[a][b] with G = <b, a> rounded to float.  The bound on |P_b - <b, r>| grows by
// 2^-21 (|c| + |r|) per update (see E below) and is part of the survivor test, so results stay bit-identical; a new pair gets
// its first approximations from 24 MFMAs on the block's filter tiles.  G is 32 132 x 32 640 floats per channel (4.2 GB, built
// on the device once per context) -- HBM capacity traded for not re-reading 16 KiB of operand tiles per pair and step.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "mp_device.h"

namespace mpc {

namespace {

constexpr int N = 64;
constexpr int kGroups = 1;                       // column groups (16 tile-channels each) per wave
constexpr int kWavesDouble = 12;                 // waves per workgroup: three per SIMD (168 VGPRs each)
constexpr int kWavesFloat = 12;                  // measured: 16 waves (128 VGPRs) spill 32 registers and run 9 % slower than 12
constexpr int kWavesMax = 16;                    // the per-wave scratch is sized for either
constexpr int kTilesLds = kBaseFilterTiles + kBlockFilterTiles;     // 36
constexpr double W_R = 0.299, W_G = 0.587, W_B = 0.114;            // ImageHelper/inc/misc.h:7-11
constexpr double U_SCALE = 0.436 / (1.0 - 0.114);
constexpr double V_SCALE = 0.615 / (1.0 - 0.299);
#ifndef MPC_SLACK
#define MPC_SLACK 0x1p-13f
#endif
constexpr float kSlack = MPC_SLACK;              // E = kSlack |r~| + kAbs bounds the split-bf16 MFMA error (derivation: DESIGN.md 3);
                                                 // a larger value stays valid (more survivors, same results): -DMPC_SLACK=... measures that
constexpr float kAbs = 0x1p-100f;
constexpr float kHuge = 1.0e30f;                 // beyond this the f32 side may overflow: the tile-channel is evaluated exhaustively
#ifndef MPC_REFILL_AT
#define MPC_REFILL_AT 16
#endif
// Finished slots of a wave are refilled from the queue once this many are free.  Measured (MI355X, 16 Mpixel frames): 1, 2, 4, 8
// and 16 all within 3 % of each other, 16 the best on the natural frame (2.57 ms vs 2.66 at 4) -- filling the slots early keeps
// 14+ of 16 busy instead of 11.6, but slots at different steps make every wave-step pay for its oldest slot's pair list.
constexpr int kRefillAt = MPC_REFILL_AT;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int COUNT, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (COUNT > 0) {
        static_for<COUNT - 1>(f);
        f(std::integral_constant<int, COUNT - 1>{});
    }
}

__device__ __forceinline__ unsigned short bf16_of(float x)          // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
{
    const __bf16 b = static_cast<__bf16>(x);
    return __builtin_bit_cast(unsigned short, b);
}

// ---- the four lanes of a slot are l, l^16, l^32, l^48: exchanges between them as VALU lane swaps (no LDS crossbar) -------
// v_permlane16_swap(x, x): [0] = x of lane - 16 in odd rows of 16, [1] = x of lane + 16 in even rows (else own x);
// v_permlane32_swap(x, x): [0] = x of lane - 32 in lanes 32..63, [1] = x of lane + 32 in lanes 0..31 (else own x).
template <class Op>
__device__ __forceinline__ unsigned reduce4_u(unsigned x, Op op)
{
    const auto a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const unsigned m = op(a[0], a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    return op(b[0], b[1]);
}
__device__ __forceinline__ float reduce4_max(float x)
{
    return __uint_as_float(reduce4_u(__float_as_uint(x), [](unsigned p, unsigned q) {
        return __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(p), __uint_as_float(q), __builtin_inff()));
    }));
}
__device__ __forceinline__ unsigned reduce4_add(unsigned x) { return reduce4_u(x, [](unsigned p, unsigned q) { return p + q; }); }
__device__ __forceinline__ double reduce4_add(double x)
{
    unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double m = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = (unsigned)__double2loint(m); hi = (unsigned)__double2hiint(m);
    const auto c = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}
// which of the two results of a swap(x, x) is the OTHER lane's value
__device__ __forceinline__ double from_prev16(double x)             // odd rows: x of lane - 16
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ double from_next16(double x)             // even rows: x of lane + 16
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double from_prev32(double x)             // lanes 32..63: x of lane - 32
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]);
}
__device__ __forceinline__ double from_next32(double x)             // lanes 0..31: x of lane + 32
{
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[1], (int)a[1]);
}

__device__ __forceinline__ float from_prev16(float x) { return __uint_as_float(__builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)[0]); }
__device__ __forceinline__ float from_next16(float x) { return __uint_as_float(__builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false)[1]); }
__device__ __forceinline__ float from_prev32(float x) { return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false)[0]); }
__device__ __forceinline__ float from_next32(float x) { return __uint_as_float(__builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false)[1]); }

// Lane row h (= lane >> 4) holds the pixels 16 * pos(h) .. + 15 with pos = 0, 1, 3, 2: the running sum of a dot product then
// travels row 0 -> 1 (16-lane swap) -> 3 (32-lane swap) -> 2 (16-lane swap), every hop one VALU lane swap.
__device__ __forceinline__ int pos_of(int h) { return h ^ (h >> 1); }

// Sum of the 64 terms held 16 per lane by the four lanes of a slot, in pixel order 0..63, starting from 0.0 (the reference's
// `tot = 0; tot += l * r`, mathmatrix.cpp:436-444); the result is returned in all four lanes.
template <class T>
__device__ __forceinline__ T chain_sum(const T (&term)[16], int h)
{
    T t = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += term[i];                      // pixels 0..15: meaningful in row 0
    T u = from_prev16(t);
#pragma unroll
    for (int i = 0; i < 16; ++i) u += term[i];                      // 16..31: row 1
    t = h == 1 ? u : t;
    u = from_prev32(t);
#pragma unroll
    for (int i = 0; i < 16; ++i) u += term[i];                      // 32..47: row 3
    t = h == 3 ? u : t;
    u = from_next16(t);
#pragma unroll
    for (int i = 0; i < 16; ++i) u += term[i];                      // 48..63: row 2 holds the whole sum
    t = h == 2 ? u : t;
    u = from_prev16(t);
    t = h == 3 ? u : t;                                             // rows 2 and 3
    u = from_next32(t);
    return h < 2 ? u : t;
}

// The two largest approximations a lane has seen, as KEYS: the value's bits with the low mantissa bits replaced by a code
// that names the row (so one v_and_or + two v_med3 track value and row together, and the runner-up's row is known as well).
// A key read as a float is >= the truncated value and < value * (1 + 2^-(23 - bits)): callers widen their thresholds by that.
// Values are finite and >= 0 (callers divert NaN / infinity / overflow before relying on it), so float order = key order.
struct TopKeys {
    float k1 = 0.0f, k2 = 0.0f;                                     // k1 >= k2
    // mask and code in registers: one v_and_or_b32; keys are non-negative floats, so the integer maximum is the float maximum
    __device__ __forceinline__ void see(unsigned value_bits, unsigned keep_mask, unsigned code)
    {
        const unsigned key = (value_bits & keep_mask) | code;
        k2 = __builtin_amdgcn_fmed3f(k1, k2, __uint_as_float(key));
        const unsigned k1b = __float_as_uint(k1);
        k1 = __uint_as_float(k1b > key ? k1b : key);
    }
};
// The mask of a key lives in a VECTOR register: v_and_or_b32 may read one scalar operand only (the code, a loop counter), so a
// scalar or literal mask would split every key into v_and + v_or -- one more VALU instruction per tracked value.
__device__ __forceinline__ unsigned in_vgpr(unsigned x)
{
    unsigned v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(x));
    return v;
}
__device__ __forceinline__ unsigned in_sgpr(unsigned x)          // x is wave-uniform; keeps the compiler from folding it into a literal
{
    unsigned v;
    asm volatile("s_mov_b32 %0, %1" : "=s"(v) : "s"(x));
    return v;
}
constexpr unsigned kKeepRow = 0x7FFFFF80u;        // |value| with 7 code bits: tile (5) and row within the lane's four (2)
constexpr unsigned kKeepPair = 0x7FFFFE00u;      // |value| with 9 code bits: pair (5), tile (2), row (2)

// Survivors of one lane waiting for their exact evaluation.  code = segment << 16 | pair << 11 | row; segment 0 = base
// rows, 1 = DetailBasis[0], 2 = pair.  More than four on one lane: the tile-channel falls back to evaluating every row.
struct Survivors {
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0, count = 0;
    bool overflow = false;
    __device__ __forceinline__ void push(int code)
    {
        if (count == 0) q0 = code; else if (count == 1) q1 = code; else if (count == 2) q2 = code; else if (count == 3) q3 = code;
        else overflow = true;
        if (count < 4) ++count;
    }
    __device__ __forceinline__ int at(int i) const { return i == 0 ? q0 : (i == 1 ? q1 : (i == 2 ? q2 : q3)); }
};

// State of one tile-channel, replicated in the four lanes of its slot.
struct TileChannel {
    int prev_id = 0;
    int next_off = 0;            // dictionary index where the next unlocked block starts (510 + rows appended so far)
    int off0 = 0;                // index of DetailBasis[0]'s first row once unlocked
    int npairs = 0;
    int fresh = -1;              // pair created by the last step: its approximations come from the MFMAs, not from G
    unsigned pk0 = 0, pk1 = 0;   // block (9 bits) | rows (7 bits) of pairs 0, 1 and 2, 3: the common case needs no meta load
    unsigned swept = 0;
    bool has0 = false, live = false;
    double coeff = 0.0;          // pending residual update (0 = none): r -= coeff * row
    int sel_g = 0;               // Gram row of the chosen atom (base row, or 510 + detail row)
    int step = 0;                // MP step this tile-channel is at (slots of a wave are refilled one by one: steps differ)
    int t = 0;                   // its index in the channel's list of tile-channels
    __device__ __forceinline__ unsigned packed(int p) const { return ((p < 2 ? pk0 : pk1) >> (16 * (p & 1))) & 0xFFFFu; }
};

__device__ __forceinline__ void load16(double (&x)[16], const double* p)
{
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2 v = q[i];
        x[2 * i] = v.x;
        x[2 * i + 1] = v.y;
    }
}

__device__ __forceinline__ void load16(float (&x)[16], const float* p)
{
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float4 v = q[i];
        x[4 * i] = v.x;
        x[4 * i + 1] = v.y;
        x[4 * i + 2] = v.z;
        x[4 * i + 3] = v.w;
    }
}

__device__ __forceinline__ double abs_of(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float abs_of(float x) { return __builtin_fabsf(x); }

template <class T>
__device__ __forceinline__ void keep_better(T& v, int& i, int& sel, T ov, int oi, int osel)
{
    const T a = abs_of(ov), b = abs_of(v);
    if (oi >= 0 && (a > b || (a == b && i >= 0 && oi < i))) {       // start: v = 0.0, i = -1: a strict '>' 0 like Select()
        v = ov; i = oi; sel = osel;
    }
}

// The lane's number, computed where it is asked for.  Lane-derived values (slot, lane row, scratch addresses) that stay live through
// the whole main loop are what the register allocator spills first -- and a spilled value comes back from scratch memory behind an
// s_waitcnt vmcnt(0), a full memory round trip in the middle of a phase.  Two VALU instructions per phase are cheaper: every phase
// of the main loop starts from FRESH_LANE instead of reading loop-long values (the asm is volatile: neither hoisted nor merged).
__device__ __forceinline__ int lane_id_fresh()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
#define FRESH_LANE                                                                    \
    const int lane = lane_id_fresh();                                                 \
    const int slot = lane & 15, h = lane >> 4;                                        \
    const int pix0 = 16 * pos_of(h);                                                  \
    (void)slot; (void)h; (void)pix0;

// In-kernel phase stamps: diagnostic builds only (-DMPC_STAMPS, tools/stamps.sh); the product build has none.
#ifdef MPC_STAMPS
#define STAMP(i)                                              \
    {                                                         \
        __builtin_amdgcn_sched_barrier(0);                    \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                    \
        stamp_acc[i] += now_ - stamp_prev;                    \
        stamp_prev = now_;                                    \
    }
#define COUNT(i, n) stamp_acc[i] += (unsigned long long)(n);
#else
#define STAMP(i)
#define COUNT(i, n)
#endif

}  // namespace

// --------------------------------------------------------------------------------------------------
// T = double: the reference's double path, bit for bit.  T = float: the `...Fast` flavour (MatchingPursuit.cpp:27-37,76-107):
// residual, exact evaluation, quantisation and update in float on the dictionary rounded to float -- the same screen (its
// bound covers the float chain's own rounding, 64 x 2^-24 of the same sum), the same survivors logic, the same order.
template <class T, int kWaves>
__global__ __launch_bounds__(64 * kWaves, kWaves / 4) void mp_pursuit_kernel(const PursuitArgs a)
{
    constexpr bool kFast = std::is_same<T, float>::value;
#ifdef MPC_STAMPS
    unsigned long long stamp_acc[24];
    for (int i = 0; i < 24; ++i) stamp_acc[i] = 0;
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long wg_begin = __builtin_amdgcn_s_memrealtime();      // 100 MHz: workgroup residency timeline
#endif
    __shared__ uint4 s_tiles[kTilesLds * 256];                      // [tile][operand q][lane], 4 KiB per tile
    __shared__ int s_rows[512], s_rowoff[512];                      // block_rows / block_row_off
    __shared__ unsigned s_touch[64 * kWaves];                       // landing zone of the cache-touch loads (never read)
    __shared__ double s_quant[3][kMaxDeviceK];                      // the quantisers (steps differ per lane, channels per wave)
    // Tiles 0 .. 31 (base rows) serve every channel; tiles 32 .. 35 hold DetailBasis[0] of ONE channel, the workgroup's "home".
    // A wave whose channel is not the home reads those four tiles from memory instead (16 KiB per step, L2-resident).  The last
    // wave to leave the home channel makes its next channel the home (see the channel switch below).
    __shared__ int s_home;                                          // home channel; -1 while tiles 32 .. 35 are being replaced
    __shared__ int s_on[3];                                         // waves of this workgroup at work on each channel
    const int ch_first = a.n_tc[0] > 0 ? 0 : (a.n_tc[1] > 0 ? 1 : 2);
    const T* const base_rows = reinterpret_cast<const T*>(kFast ? static_cast<const void*>(a.base32) : static_cast<const void*>(a.base));
    {
        const uint16_t* const blk0 = ch_first == 0 ? a.block_tiles[0] : (ch_first == 1 ? a.block_tiles[1] : a.block_tiles[2]);
        const uint4* src_base = reinterpret_cast<const uint4*>(a.base_tiles);
        const uint4* src_blk0 = reinterpret_cast<const uint4*>(blk0);               // block 0 = the first four tiles
        for (int i = threadIdx.x; i < kTilesLds * 256; i += 64 * kWaves)
            s_tiles[i] = i < kBaseFilterTiles * 256 ? src_base[i] : src_blk0[i - kBaseFilterTiles * 256];
        for (int i = threadIdx.x; i < 512; i += 64 * kWaves) {
            s_rows[i] = i < a.num_base ? a.block_rows[i] : 0;
            s_rowoff[i] = i < a.num_base ? a.block_row_off[i] : 0;
        }
        for (int i = threadIdx.x; i < 3 * kMaxDeviceK; i += 64 * kWaves)
            s_quant[i / kMaxDeviceK][i % kMaxDeviceK] = (i % kMaxDeviceK) < a.K ? a.quant[(i / kMaxDeviceK) * a.K + i % kMaxDeviceK] : 1.0;
        if (threadIdx.x == 0) {
            s_home = ch_first;
            s_on[0] = s_on[1] = s_on[2] = 0;
            s_on[ch_first] = kWaves;
        }
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // scalar: so are the wave's scratch pointers.  lane, slot (= lane & 15), lane row h (= lane >> 4) and pix0 (the first of
                                                                    // the lane's 16 consecutive pixels): FRESH_LANE, phase by phase
    // the wave's channel and everything that depends on it (wave-uniform: scalar registers)
    int ch = ch_first;
    const T* detail = nullptr;
    const uint16_t* block_tiles = nullptr;
    const float* gram = nullptr;
    unsigned* queue = nullptr;
    long long n_tc = 0;
    auto bind_channel = [&](int c) {
        c = __builtin_amdgcn_readfirstlane(c);
        ch = c;
        detail = reinterpret_cast<const T*>(kFast ? static_cast<const void*>(c == 0 ? a.detail32[0] : (c == 1 ? a.detail32[1] : a.detail32[2]))
                                                  : static_cast<const void*>(c == 0 ? a.detail[0] : (c == 1 ? a.detail[1] : a.detail[2])));
        block_tiles = c == 0 ? a.block_tiles[0] : (c == 1 ? a.block_tiles[1] : a.block_tiles[2]);
        gram = c == 0 ? a.gram[0] : (c == 1 ? a.gram[1] : a.gram[2]);
        queue = a.queue + c;
        n_tc = c == 0 ? a.n_tc[0] : (c == 1 ? a.n_tc[1] : a.n_tc[2]);
    };
    bind_channel(ch_first);
    const long long wave_global = (long long)blockIdx.x * kWaves + wave;
    float* const my_p = a.pair_p + wave_global * (kGroups * 16 * kMaxPairs * 64);
    unsigned* const my_meta = a.pair_meta + wave_global * (kGroups * 16 * kMaxPairs * 2);
    float* const my_e = a.pair_e + wave_global * (kGroups * 16 * kMaxPairs);
    unsigned* const my_touch = s_touch + 64 * wave;
    const int K = a.K;

    const unsigned keep_row_mask = in_vgpr(kKeepRow), keep_pair_mask = in_vgpr(kKeepPair);
    unsigned n_mfma = 0, n_steps = 0;                               // executed MFMA instructions / tile-channel-steps of this wave
    int unit[kGroups];                                              // >= 0: a slot of the group is live
    T r[kGroups][16];
    TileChannel tc[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
        unit[g] = -1;
#pragma unroll
        for (int i = 0; i < 16; ++i) r[g][i] = 0;
    }
    bool queue_empty = false;

    auto tile_mfma = [&](const uint4 (&av)[4], const bf16x8 (&hi)[2], const bf16x8 (&lo)[2]) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 ah, al;
            __builtin_memcpy(&ah, &av[2 * kk + 0], 16);
            __builtin_memcpy(&al, &av[2 * kk + 1], 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, hi[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, lo[kk], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, hi[kk], acc, 0, 0, 0);
        }
        return acc;
    };
    // record index of tile-channel t of this channel: counts[rec], choices[rec * K + step]
    auto rec_of = [&](int t) { return a.vec_in ? (long long)t : (long long)t * 3 + ch; };
    auto lds_tile = [&](uint4 (&dst)[4], int t, int lane) {
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = s_tiles[t * 256 + q * 64 + lane];
    };
    // tile t of DetailBasis[0] of the wave's channel from memory (a wave away from the workgroup's home channel)
    auto far_tile = [&](uint4 (&dst)[4], int t, int lane) {
        const uint4* tiles = reinterpret_cast<const uint4*>(block_tiles) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = tiles[t * 256 + q * 64];
    };

    // ---- pairs: P_b -= float(c) * G[a][b] for the atom a chosen by the step just finished; top two upper bounds |P_b| + E as
    //      keys.  The update of a step's pairs needs only that step's choice (atom, coefficient), not the updated residual:
    //      it runs at the END of the step, between the request for the chosen row and the residual update that waits for it,
    //      and its results (tp, lbmax, oddp) are used by the next step's thresholds.  rbound = |r~| of the residual BEFORE the
    //      update bounds the updated one as well (an MP step with a non-zero quantised coefficient does not lengthen it, up
    //      to rounding covered by the 1.0000002 in it): a valid, marginally wider E.
    TopKeys tp[kGroups];
    float lbmax[kGroups];
    bool oddp[kGroups];
    float rbound[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) { lbmax[g] = -3.0e38f; oddp[g] = false; rbound[g] = 0.0f; }
    // same_p: p is the same in every lane (a loop counter): every key's code then sits in a scalar register of its own
    auto pair_update = [&](auto gc, auto same_p, int p, bool upd, unsigned info, float4 (&pv)[4], const float4 (&gv)[4], float E) {
        constexpr int g = decltype(gc)::value;
        FRESH_LANE
        const long long pi = (long long)(g * 16 + slot) * kMaxPairs + p;
        const int rows = (int)((info >> 9) & 127u);
        if (upd) {
            const float c32 = (float)tc[g].coeff;
            float4* pp = reinterpret_cast<float4*>(my_p + pi * 64 + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                pv[t].x = fmaf(-c32, gv[t].x, pv[t].x);
                pv[t].y = fmaf(-c32, gv[t].y, pv[t].y);
                pv[t].z = fmaf(-c32, gv[t].z, pv[t].z);
                pv[t].w = fmaf(-c32, gv[t].w, pv[t].w);
                pp[4 * t] = pv[t];
            }
            // |P_b - <b, r>| grows by at most 2^-23 |c| (rounding of c and of G) + 2^-24 (|r| + E) (the fma's result)
            E = (E + 0x1p-21f * (fabsf(c32) + rbound[g])) * 1.000001f;
            if (h == 0) my_e[pi] = E;
            oddp[g] = oddp[g] || !(fabsf(c32) < kHuge);
        }
        oddp[g] = oddp[g] || !(E < kHuge);
        // rows 62 and 63 of a block can be pads (blocks have 62 or 63 rows): lane row h = 3, tile 3, v = 2, 3.  Their P and G
        // are exactly 0; their upper bound must be 0 too (not E), or a pad could pass for a survivor.
        const bool pad2 = h == 3 && rows < 63, pad3 = h == 3 && rows < 64;
        const unsigned keepp = keep_pair_mask;
        float biggest = 0.0f;                                   // of this pair's |P|: its lower bound is biggest - E
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float vals[4] = {pv[t].x, pv[t].y, pv[t].z, pv[t].w};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float m = fabsf(vals[v]);
                float ub = m + E;
                if (t == 3 && v == 2) ub = pad2 ? 0.0f : ub;
                if (t == 3 && v == 3) ub = pad3 ? 0.0f : ub;
                const unsigned code = (unsigned)((p << 4) | (t << 2) | v);
                tp[g].see(__float_as_uint(ub), keepp, decltype(same_p)::value ? in_sgpr((unsigned)__builtin_amdgcn_readfirstlane((int)code)) : code);
                biggest = __builtin_amdgcn_fmed3f(biggest, m, __builtin_inff());
            }
        }
        lbmax[g] = __builtin_amdgcn_fmed3f(lbmax[g], biggest - E, __builtin_inff());
    };
    for (;;) {
        // ---- (1) refill: slots whose tile-channel has ended take the next tile-channels from the queue, kRefillAt or more at
        //      a time (one atomic per refill; a wave whose slots are all free refills at once)
        bool any_unit = false;
        for (;;) {
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (queue_empty) return;
            const unsigned free_slots = (unsigned)(__ballot(!tc[g].live) & 0xFFFFull);      // lane row 0 speaks for its slot
            const int n_free = __popc(free_slots);
            if (!(n_free >= kRefillAt || (unit[g] < 0 && n_free > 0))) return;
            FRESH_LANE
            int first = 0;
            if (lane == 0) first = (int)atomicAdd(queue, (unsigned)n_free);
            first = __builtin_amdgcn_readfirstlane(first);
            if ((long long)first + n_free >= n_tc) queue_empty = true;
            const long long t = (long long)first + __popc(free_slots & ((1u << slot) - 1u));
            const bool take = !tc[g].live && t < n_tc;
            if (!__ballot(take)) return;
            unit[g] = 0;
            TileChannel s;
            s.live = true;
            s.t = (int)t;
            s.next_off = a.num_base;
            const long long ts = take ? t : 0;                      // loads below are unconditional: always a valid address
            T fresh_r[16];
            if (a.vec_in) {
                double in[16];
                load16(in, a.vec_in + ts * N + pix0);
#pragma unroll
                for (int i = 0; i < 16; ++i) fresh_r[i] = (T)in[i];
            } else {
                const int tiles_per_frame = a.tiles_x * a.tile_rows;
                const int frame = (int)(ts / tiles_per_frame);
                const int tile = (int)(ts - (long long)frame * tiles_per_frame);
                const int tx = tile / a.tile_rows, ty = a.tile_row_begin + (tile - tx * a.tile_rows);
                if (a.out_tile_rows > 0) s.t = (frame * a.tiles_x + tx) * a.out_tile_rows + ty;      // the records' place in a whole frame
                const uint8_t* img = a.rgb + (long long)frame * a.frame_stride;
                unsigned char px[16][3];
                // the lane's 16 pixels are two rows of 8 (24 bytes each, at 24 tx: 8-byte aligned when the rows are).  Whole tiles
                // of an 8-byte aligned image take them as six 8-byte loads; ragged edges and odd strides byte by byte, clamped
                const bool whole = tx * 8 + 8 <= a.width && ty * 8 + 8 <= a.height;
                if (a.rgb_aligned8 && !__ballot(!whole)) {
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr) {
                        const uint2* p = reinterpret_cast<const uint2*>(img + (long long)(ty * 8 + (pix0 >> 3) + rr) * a.row_stride + 24 * tx);
                        const uint2 v0 = p[0], v1 = p[1], v2 = p[2];
                        const unsigned w[6] = {v0.x, v0.y, v1.x, v1.y, v2.x, v2.y};
#pragma unroll
                        for (int b = 0; b < 24; ++b) px[8 * rr + b / 3][b % 3] = (unsigned char)((w[b >> 2] >> (8 * (b & 3))) & 0xFFu);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {                  // block index dx + 8 dy = pix0 + i
                        const int x = tx * 8 + (i & 7), y = ty * 8 + (pix0 >> 3) + (i >> 3);
                        const int xc = x < a.width ? x : a.width - 1, yc = y < a.height ? y : a.height - 1;
                        const uint8_t* p = img + (long long)yc * a.row_stride + 3 * xc;
                        px[i][0] = p[0]; px[i][1] = p[1]; px[i][2] = p[2];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int x = tx * 8 + (i & 7), y = ty * 8 + (pix0 >> 3) + (i >> 3);
                    const double red = (double)px[i][0], green = (double)px[i][1], blue = (double)px[i][2];
                    const double Y = (W_R * red + W_G * green + W_B * blue);            // misc.cpp:12-21, same expression order
                    const double v = ch == 0 ? Y : (ch == 1 ? (U_SCALE * (blue - Y)) : (V_SCALE * (red - Y)));
                    fresh_r[i] = (T)((x < a.width && y < a.height) ? v : 0.0);         // zero fill (CompressedImage.cpp:548-552); Fast: static_cast<float> (:601-605)
                }
            }
            if (take) {
#pragma unroll
                for (int i = 0; i < 16; ++i) r[g][i] = fresh_r[i];
                tc[g] = s;
            }
        });
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any_unit = any_unit || unit[g] >= 0;
        if (any_unit || !queue_empty) break;
        // ---- channel switch: the wave holds nothing and its channel's queue is dry -> the next channel that has work.  The
        //      workgroup's LDS holds DetailBasis[0] of its home channel only: the last wave to leave the home channel (no wave
        //      reads tiles 32 .. 35 any more) copies its next channel's four tiles there and declares it the new home; until then,
        //      and for waves on yet another channel, those tiles come from memory (`home` below).
        int next = ch + 1;
        while (next < 3 && (next == 1 ? a.n_tc[1] : a.n_tc[2]) == 0) ++next;
        if (next >= 3) break;
        int was_on = 0;
        FRESH_LANE
        if (lane == 0) {
            __hip_atomic_fetch_add(&s_on[next], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            was_on = __hip_atomic_fetch_add(&s_on[ch], -1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        was_on = __builtin_amdgcn_readfirstlane(was_on);
        const int old_home = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_home, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (was_on == 1 && old_home == ch) {
            if (lane == 0) __hip_atomic_store(&s_home, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const uint4* src = reinterpret_cast<const uint4*>(next == 1 ? a.block_tiles[1] : a.block_tiles[2]);
#pragma unroll 4
            for (int i = lane; i < kBlockFilterTiles * 256; i += 64) s_tiles[kBaseFilterTiles * 256 + i] = src[i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&s_home, next, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        bind_channel(next);
        queue_empty = false;
        }
        if (!any_unit) break;
        // does this workgroup's LDS hold DetailBasis[0] of this wave's channel?  (it cannot change while the wave has live slots)
        const bool home = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&s_home, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) == ch;
        static_for<kGroups>([&](auto gc) { n_steps += (unsigned)__popcll(__ballot(unit[decltype(gc)::value] >= 0 && tc[decltype(gc)::value].live) & 0xFFFFull); });
        STAMP(0)
        COUNT(12, 1)
        static_for<kGroups>([&](auto gc) { COUNT(16, __popcll(__ballot(tc[decltype(gc)::value].live && unit[decltype(gc)::value] >= 0))) });

        // ---- (2a) B operands from the residuals; error bound of this step's MFMA approximations ----------------------
        bf16x8 bh[kGroups][2], bl[kGroups][2];
        float Eb[kGroups], rnorm[kGroups];
        bool nz[kGroups], odd[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            double ss = 0.0;
            bool nzl = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const T d = r[g][i];
                nzl = nzl || (d != 0);
                const float x = (float)d;
                ss += (double)x * (double)x;
                const unsigned short hb = bf16_of(x);
                const unsigned short lb = bf16_of(x - __uint_as_float((unsigned)hb << 16));
                bh[g][i >> 3][i & 7] = (short)hb;
                bl[g][i >> 3][i & 7] = (short)lb;
            }
            ss = reduce4_add(ss);
            nz[g] = reduce4_add(nzl ? 1u : 0u) != 0u;
            rnorm[g] = (float)__builtin_sqrt(ss) * 1.0000002f;
            Eb[g] = kSlack * rnorm[g] + kAbs;
            rbound[g] = rnorm[g];
            odd[g] = !(rnorm[g] < kHuge);                           // NaN, infinity or large enough to overflow the f32 side
        });
        STAMP(1)

        // ---- (2c) a pair created by the last step: first approximations of its 64 rows from the block's filter tiles (their
        //      lines were pulled towards the cache when the pair was created)
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0) return;
            const bool mine = tc[g].live && tc[g].fresh >= 0;
            unsigned long long pend = __ballot(mine) & 0xFFFFull;       // lane row 0 speaks for its slot
            if (!pend) return;
            FRESH_LANE
            COUNT(19, __popcll(pend))
            unsigned fresh_info = 0;
            if (mine)
                fresh_info = (tc[g].fresh < 4 ? tc[g].packed(tc[g].fresh) : my_meta[2 * ((long long)(g * 16 + slot) * kMaxPairs + tc[g].fresh)]) & 0xFFFFu;
            float4 pnew[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pnew[t] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            while (pend) {
                const int src = __builtin_ctzll(pend);
                pend &= pend - 1;
                const int blk = __builtin_amdgcn_readlane((int)(fresh_info & 511u), src);
                n_mfma += 6 * kBlockFilterTiles;
                const uint4* tiles = reinterpret_cast<const uint4*>(block_tiles) + (long long)blk * (kBlockFilterTiles * 256) + lane;
#ifndef MPC_NEWPAIR_TILES
#define MPC_NEWPAIR_TILES 2
#endif
                // tiles (4 KiB each) requested before the first is used.  Measured: all four at once (one round trip instead of
                // two) is 5 % SLOWER overall -- 32 more live registers at this point cost more elsewhere than the round trip saves
                constexpr int kInFlight = MPC_NEWPAIR_TILES;
#pragma unroll
                for (int t2 = 0; t2 < kBlockFilterTiles; t2 += kInFlight) {
                    uint4 av[kInFlight][4];
#pragma unroll
                    for (int t = 0; t < kInFlight; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) av[t][q] = tiles[(t2 + t) * 256 + q * 64];
#pragma unroll
                    for (int t = 0; t < kInFlight; ++t) {
                        const f32x4 acc = tile_mfma(av[t], bh[g], bl[g]);
                        if (slot == src) pnew[t2 + t] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    }
                }
            }
            if (mine) {
                float4* dst = reinterpret_cast<float4*>(my_p + ((long long)(g * 16 + slot) * kMaxPairs + tc[g].fresh) * 64 + 4 * h);
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[4 * t] = pnew[t];
                if (h == 0) my_e[(g * 16 + slot) * kMaxPairs + tc[g].fresh] = Eb[g];
                const float4 none[4] = {};
                pair_update(gc, std::false_type{}, tc[g].fresh, false, fresh_info, pnew, none, Eb[g]);
            }
        });
        STAMP(3)

        // ---- (3) pass 1 over the LDS tiles: top two approximations per lane as keys, base rows and DetailBasis[0] apart.  Both
        //      groups always run (an empty group computes on stale registers and is ignored): no branch between the MFMAs of a
        //      tile and the tracking of the previous tile's results, which fills their shadow.  No global memory in here: the
        //      stores of the pair updates drain meanwhile.
        TopKeys tb[kGroups], td[kGroups];
        bool any0 = false;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any0 = any0 || (unit[g] >= 0 && __ballot(tc[g].live && tc[g].has0) != 0);
        n_mfma += 6 * kGroups * (kBaseFilterTiles + (any0 ? kBlockFilterTiles : 0));
        {
            FRESH_LANE
            uint4 ta[4];
            f32x4 acc[kGroups], prev[kGroups];
            const unsigned keep = keep_row_mask;
            auto track = [&](TopKeys (&trk)[kGroups], const f32x4 (&val)[kGroups], unsigned code0) {
                unsigned code[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) code[v] = (unsigned)__builtin_amdgcn_readfirstlane((int)(code0 + v));    // each in a scalar register
#pragma unroll
                for (int g = 0; g < kGroups; ++g)
#pragma unroll
                    for (int v = 0; v < 4; ++v) trk[g].see(__float_as_uint(val[g][v]), keep, code[v]);
            };
            // ONE tile buffer: a tile's six MFMAs have read it by the time the next tile's LDS reads land (they are issued
            // behind the last MFMA and return while the matrix pipe works and the previous tile's results are tracked); a second
            // buffer bought nothing but 16 more live registers in the phase with the most of them
            lds_tile(ta, 0, lane);
#pragma unroll
            for (int g = 0; g < kGroups; ++g) prev[g] = tile_mfma(ta, bh[g], bl[g]);
#pragma unroll 1
            for (int t = 1; t < kBaseFilterTiles - 1; t += 2) {     // results alternate between acc and prev: no register copies
                lds_tile(ta, t, lane);
#pragma unroll
                for (int g = 0; g < kGroups; ++g) acc[g] = tile_mfma(ta, bh[g], bl[g]);
                track(tb, prev, 4 * (t - 1));
                lds_tile(ta, t + 1, lane);
#pragma unroll
                for (int g = 0; g < kGroups; ++g) prev[g] = tile_mfma(ta, bh[g], bl[g]);
                track(tb, acc, 4 * t);
            }
            // tiles 0 .. 30 are done or in `prev` (tile 30)
            lds_tile(ta, kBaseFilterTiles - 1, lane);
#pragma unroll
            for (int g = 0; g < kGroups; ++g) acc[g] = tile_mfma(ta, bh[g], bl[g]);
            track(tb, prev, 4 * (kBaseFilterTiles - 2));
            track(tb, acc, 4 * (kBaseFilterTiles - 1));
            if (any0) {
#pragma unroll 1
                for (int t = 0; t < kBlockFilterTiles; ++t) {
                    if (home) lds_tile(ta, kBaseFilterTiles + t, lane);
                    else far_tile(ta, t, lane);
#pragma unroll
                    for (int g = 0; g < kGroups; ++g) acc[g] = tile_mfma(ta, bh[g], bl[g]);
                    track(td, acc, 4 * t);
                }
            }
        }
        STAMP(4)

        // ---- (5) thresholds and survivors --------------------------------------------------------------------------------
        // Every row's exact |p| lies in [|a| - E, |a| + E] (E = Eb for MFMA rows, the pair's own bound otherwise).  With
        // T = the largest lower bound, the row Select() returns -- and every row tying with it -- has |a| + E >= T.  A key
        // stands for a value in [key & mask, key * (1 + 2^-16)] (2^-14 for pairs): thresholds on keys are lowered accordingly.
        // A lane whose runner-up key also reaches the threshold may hold more than two survivors: pass 2 / rescan lists them.
        Survivors sv[kGroups];
        float thr_b[kGroups], Tmin[kGroups];
        bool pass2[kGroups], rescan[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            pass2[g] = false; rescan[g] = false; thr_b[g] = 0.0f; Tmin[g] = 0.0f;
            if (unit[g] < 0) return;
            FRESH_LANE
            const bool has0 = tc[g].has0, hasp = tc[g].npairs > 0;
            const float top_b = __uint_as_float(__float_as_uint(reduce4_max(tb[g].k1)) & kKeepRow);
            const float top_d = __uint_as_float(__float_as_uint(reduce4_max(td[g].k1)) & kKeepRow);
            const float lbm = reduce4_max(lbmax[g]);
            const bool bad = odd[g] || (hasp && reduce4_add(oddp[g] ? 1u : 0u) != 0u);
            float t = top_b - Eb[g];
            if (has0) t = fmaxf(t, top_d - Eb[g]);
            if (hasp) t = fmaxf(t, lbm);
            Tmin[g] = bad ? -__builtin_inff() : t;                     // every row survives: the exhaustive evaluation takes over
            thr_b[g] = (Tmin[g] - Eb[g]) * (1.0f - 0x1p-15f);          // for MFMA rows, on keys or values
            const float thr_p = Tmin[g] * (1.0f - 0x1p-13f);           // for the pairs' upper-bound keys
            const bool live = tc[g].live && nz[g];                 // an all-zero residual projects to 0 everywhere: index -1
            const bool vague = bad || !(thr_b[g] > 0.0f);           // zero pads and everything else would qualify
            const bool unclear_b = vague || !(tb[g].k2 < thr_b[g]) || (has0 && !(td[g].k2 < thr_b[g]));
            const bool unclear_p = hasp && (vague || !(tp[g].k2 < thr_p));
            pass2[g] = __ballot(live && unclear_b) != 0;
            rescan[g] = live && reduce4_add(unclear_p ? 1u : 0u) != 0u;
            if (live && !pass2[g]) {
                if (!(tb[g].k1 < thr_b[g])) {
                    const unsigned code = __float_as_uint(tb[g].k1) & 127u;
                    sv[g].push((int)(16 * (code >> 2) + 4 * h + (code & 3u)));
                }
                if (has0 && !(td[g].k1 < thr_b[g])) {
                    const unsigned code = __float_as_uint(td[g].k1) & 127u;
                    sv[g].push((1 << 16) | (int)(16 * (code >> 2) + 4 * h + (code & 3u)));
                }
            }
            if (live && hasp && !rescan[g] && !(tp[g].k1 < thr_p)) {
                const unsigned code = __float_as_uint(tp[g].k1) & 511u;
                sv[g].push((2 << 16) | (int)((code >> 4) << 11) | (int)(16 * ((code >> 2) & 3u) + 4 * h + (code & 3u)));
            }
        });
        STAMP(5)

        // ---- (6) pass 2 for groups with an unclear tile-channel: the same MFMAs again, every row at or above the threshold
        bool any2 = false;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any2 = any2 || pass2[g];
        if (any2) {
            static_for<kGroups>([&](auto gc) { COUNT(13, pass2[decltype(gc)::value] ? 1 : 0) });
            FRESH_LANE
            uint4 cur[4], nxt[4];
            lds_tile(cur, 0, lane);
            const int last = any0 ? kTilesLds : kBaseFilterTiles;
            auto any_tile = [&](uint4 (&dst)[4], int t) {
                if (t < kBaseFilterTiles || home) lds_tile(dst, t, lane);
                else far_tile(dst, t - kBaseFilterTiles, lane);
            };
            static_for<kGroups>([&](auto gc) { n_mfma += pass2[decltype(gc)::value] ? 6 * last : 0; });
#pragma unroll 1
            for (int t = 0; t < last; ++t) {
                any_tile(nxt, t + 1 < last ? t + 1 : t);
                const bool is_base = t < kBaseFilterTiles;
                const int row0 = (is_base ? t : t - kBaseFilterTiles) * 16 + 4 * h;
                const int limit = is_base ? a.num_base : a.rows0;
                static_for<kGroups>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if (!pass2[g]) return;
                    const f32x4 acc = tile_mfma(cur, bh[g], bl[g]);
                    const bool wanted = tc[g].live && nz[g] && (is_base || tc[g].has0);
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (wanted && !(fabsf(acc[v]) < thr_b[g]) && row0 + v < limit) sv[g].push((is_base ? 0 : (1 << 16)) | (row0 + v));
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            }
        }
        STAMP(6)
        // pairs of an unclear tile-channel: every row whose upper bound reaches T
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0 || !__ballot(rescan[g])) return;
            FRESH_LANE
            for (int p = 0; __ballot(rescan[g] && p < tc[g].npairs) != 0; ++p) {
                if (!(rescan[g] && p < tc[g].npairs)) continue;
                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + p;
                const int rows = (int)((my_meta[2 * pi] >> 9) & 127u);
                const float E = my_e[pi];
                const float4* pp = reinterpret_cast<const float4*>(my_p + pi * 64 + 4 * h);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float4 x = pp[4 * t];
                    const float vals[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * t + 4 * h + v;
                        if (row < rows && !(fabsf(vals[v]) + E < Tmin[g])) sv[g].push((2 << 16) | (p << 11) | row);
                    }
                }
            }
        });
        STAMP(7)

        // ---- (7) exact evaluation of the survivors, one per tile-channel and round, by the four lanes of its slot; both groups'
        //      rows are in flight together and their two sums run side by side
        T best_val[kGroups];
        int best_idx[kGroups], best_sel[kGroups];
        int c0[kGroups], c1[kGroups], c2[kGroups], total[kGroups];
        bool exhaustive[kGroups];
        const int lane7 = lane_id_fresh();                             // this phase's lane values (its lambdas use them)
        const int slot7 = lane7 & 15, h7 = lane7 >> 4, pix7 = 16 * pos_of(h7);
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            best_val[g] = 0; best_idx[g] = -1; best_sel[g] = 0;
            c0[g] = __shfl(sv[g].count, slot7);
            c1[g] = __shfl(sv[g].count, slot7 + 16);
            c2[g] = __shfl(sv[g].count, slot7 + 32);
            total[g] = c0[g] + c1[g] + c2[g] + __shfl(sv[g].count, slot7 + 48);
            exhaustive[g] = reduce4_add(sv[g].overflow ? 1u : 0u) != 0u;
        }
        // what a code means for this lane's tile-channel: the row's address, its dictionary index and its Gram row
        auto resolve = [&](auto gc, int code, const T*& ptr, int& idx, int& sel) {
            constexpr int g = decltype(gc)::value;
            const int seg = code >> 16, row = code & 2047;
            if (seg == 0) { ptr = base_rows + (long long)row * N; idx = row; sel = row; }
            else if (seg == 1) { ptr = detail + (long long)row * N; idx = tc[g].off0 + row; sel = a.num_base + row; }
            else {
                const int p = (code >> 11) & 31;
                const long long pi = (long long)(g * 16 + slot7) * kMaxPairs + p;
                const unsigned meta = my_meta[2 * pi];
                const int drow = s_rowoff[meta & 511u] + row;
                ptr = detail + (long long)drow * N;
                idx = (int)(meta >> 16) + row;
                sel = a.num_base + drow;
            }
        };
        auto evaluate2 = [&](const bool (&on)[kGroups], const int (&code)[kGroups]) {
            const T* ptr[kGroups];
            int idx[kGroups], sel[kGroups];
            T x[kGroups][16];
            static_for<kGroups>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                ptr[g] = base_rows; idx[g] = -1; sel[g] = 0;
                if (on[g]) resolve(gc, code[g], ptr[g], idx[g], sel[g]);
            });
#pragma unroll
            for (int g = 0; g < kGroups; ++g) load16(x[g], ptr[g] + pix7);
#pragma unroll
            for (int g = 0; g < kGroups; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[g][i] = x[g][i] * r[g][i];          // the reference's l * r, rounded to double
            T p[kGroups];
#pragma unroll
            for (int g = 0; g < kGroups; ++g) p[g] = chain_sum(x[g], h7);
#pragma unroll
            for (int g = 0; g < kGroups; ++g)
                if (on[g]) keep_better(best_val[g], best_idx[g], best_sel[g], p[g], idx[g], sel[g]);
        };
        for (int k = 0;; ++k) {
            bool on[kGroups];
            int code[kGroups];
            bool any = false;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                on[g] = unit[g] >= 0 && tc[g].live && !exhaustive[g] && k < total[g];
                any = any || on[g];
                const int owner = k < c0[g] ? 0 : (k < c0[g] + c1[g] ? 1 : (k < c0[g] + c1[g] + c2[g] ? 2 : 3));
                const int local = k - (owner == 0 ? 0 : (owner == 1 ? c0[g] : (owner == 2 ? c0[g] + c1[g] : c0[g] + c1[g] + c2[g])));
                code[g] = __shfl(sv[g].at(local & 3), slot7 + 16 * owner);
            }
            if (!__ballot(any)) break;
            COUNT(14, 1)
            evaluate2(on, code);
        }
        STAMP(8)
        // a tile-channel whose survivors did not fit (many exact ties, NaN): every row it can choose from, in order
        {
            bool ex[kGroups];
            int n_all[kGroups];
            bool any_ex = false;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                ex[g] = unit[g] >= 0 && tc[g].live && nz[g] && exhaustive[g];
                n_all[g] = a.num_base + 64 + 64 * tc[g].npairs;
                any_ex = any_ex || ex[g];
            }
            if (__ballot(any_ex)) {
                COUNT(15, 1)
                for (int n = 0;; ++n) {
                    bool on[kGroups];
                    int code[kGroups];
                    bool more = false, any = false;
                    static_for<kGroups>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        on[g] = ex[g] && n < n_all[g];
                        more = more || on[g];
                        code[g] = 0;
                        if (n < a.num_base) code[g] = n;
                        else if (n < a.num_base + 64) { code[g] = (1 << 16) | (n - a.num_base); on[g] = on[g] && tc[g].has0 && n - a.num_base < a.rows0; }
                        else {
                            const int p = (n - a.num_base - 64) >> 6, row = (n - a.num_base - 64) & 63;
                            code[g] = (2 << 16) | (p << 11) | row;
                            if (on[g]) on[g] = row < (int)((my_meta[2 * ((long long)(g * 16 + slot7) * kMaxPairs + p)] >> 9) & 127u);
                        }
                        any = any || on[g];
                    });
                    if (!__ballot(more)) break;
                    if (!__ballot(any)) continue;
                    evaluate2(on, code);
                }
            }
        }
        STAMP(9)

        // ---- (8) finish.  The chosen rows are requested first (the residual update needs them), then everything that stores.
        T xrow[kGroups][16];
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            FRESH_LANE
            const int sg = (unit[g] >= 0 && tc[g].live && best_idx[g] >= 0) ? best_sel[g] : 0;      // Gram row -> the row itself
            load16(xrow[g], (sg < a.num_base ? base_rows + (long long)sg * N : detail + (long long)(sg - a.num_base) * N) + pix0);
        }
        // delta / zig-zag, quantise, record, unlock, termination (MatchingPursuit.cpp:50-71)
        bool ended[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            ended[g] = false;
            if (unit[g] < 0) return;
            FRESH_LANE
            TileChannel& s = tc[g];
            s.coeff = 0.0;
            s.fresh = -1;
            if (!s.live) return;
            s.swept += (unsigned)s.next_off;                       // rows of this step's dictionary (SURVEY 8d "S")
            int count = 0;
            unsigned record = 0;
            if (best_idx[g] < 0) {
                ended[g] = true; count = s.step;
            } else {
                const int id = best_idx[g];
                const unsigned delta = s.step > 0 ? (((unsigned)(id - s.prev_id) << 1) ^ (unsigned)((id - s.prev_id) >> 31)) : (unsigned)id;
                s.prev_id = id;
                const T qstep = (T)s_quant[ch][s.step];             // Fast: Eigen::VectorXf quantization
                const T ratio = best_val[g] / qstep;
                const int q = kFast ? (int)__builtin_roundf((float)ratio) : (int)__builtin_round((double)ratio);
                const unsigned zz = ((unsigned)q << 1) ^ (unsigned)(q >> 31);
                record = (delta & 0xFFFFu) | ((zz & 0xFFFFu) << 16);
                if (q == 0) {
                    ended[g] = true; count = s.step;
                } else {
                    s.coeff = (double)(qstep * (T)q);              // a float product is exact in the double that carries it
                    s.sel_g = best_sel[g];
                    if (s.step + 1 == K) { ended[g] = true; count = K; }
                    else if (id < a.num_base) {                     // unlock DetailBasis[id]; a repeat is indexed, not swept again
                        const int rows = s_rows[id];
                        bool repeat = id == 0 && s.has0;
                        for (int p = 0; p < s.npairs; ++p) {
                            const unsigned blk = p < 4 ? (s.packed(p) & 511u) : (my_meta[2 * ((long long)(g * 16 + slot) * kMaxPairs + p)] & 511u);
                            repeat = repeat || (int)blk == id;
                        }
                        if (!repeat) {
                            if (id == 0) { s.has0 = true; s.off0 = s.next_off; }
                            else {
                                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + s.npairs;
                                const unsigned info = (unsigned)id | ((unsigned)rows << 9);
                                if (h == 0) my_meta[2 * pi] = info | ((unsigned)s.next_off << 16);
                                if (s.npairs < 2) s.pk0 |= info << (16 * s.npairs);
                                else if (s.npairs < 4) s.pk1 |= info << (16 * (s.npairs - 2));
                                s.fresh = s.npairs;
                                s.npairs += 1;
                            }
                        }
                        s.next_off += rows;
                    }
                }
            }
            if (h == 0) {
                const long long rec = rec_of(s.t);
                a.out.choices[rec * K + s.step] = record;
                if (ended[g]) {
                    a.out.counts[rec] = (uint16_t)count;
                    if (a.out.swept) a.out.swept[rec] = s.swept;
                }
            }
        });
        STAMP(10)

        // ---- (8b) the pairs of the tile-channels that go on, for the NEXT step (see `pair_update`): pair 0's P, G and E first
        float4 pv0[kGroups][4], gv0[kGroups][4];
        float E0[kGroups];
        bool on0[kGroups], upd0[kGroups];
        unsigned info0[kGroups];
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            FRESH_LANE
            on0[g] = unit[g] >= 0 && tc[g].live && !ended[g] && tc[g].npairs > 0 && tc[g].fresh != 0;
            upd0[g] = on0[g] && tc[g].coeff != 0.0;
            info0[g] = tc[g].packed(0);
            const long long pi = (long long)(g * 16 + slot) * kMaxPairs;
            const float4* pp = reinterpret_cast<const float4*>(my_p + pi * 64 + 4 * h);
            const float4* gp = reinterpret_cast<const float4*>(gram + (upd0[g] ? (long long)tc[g].sel_g * a.gram_stride + (int)(info0[g] & 511u) * 64 : 0) + 4 * h);
#pragma unroll
            for (int t = 0; t < 4; ++t) { pv0[g][t] = pp[4 * t]; gv0[g][t] = gp[4 * t]; }
            E0[g] = my_e[pi];
        }
#pragma unroll
        for (int g = 0; g < kGroups; ++g) { tp[g] = TopKeys(); lbmax[g] = -3.0e38f; oddp[g] = false; }
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (__ballot(on0[g])) { COUNT(17, 1) COUNT(18, __popcll(__ballot(on0[g]) & 0xFFFFull)) }
            if (on0[g]) pair_update(gc, std::true_type{}, 0, upd0[g], info0[g], pv0[g], gv0[g], E0[g]);
        });
#pragma unroll 1
        for (int p = 1;; ++p) {                                     // further pairs: both groups' loads of a round issued together
            bool on[kGroups];
            bool any = false;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                on[g] = unit[g] >= 0 && tc[g].live && !ended[g] && p < tc[g].npairs && p != tc[g].fresh;
                any = any || (unit[g] >= 0 && tc[g].live && !ended[g] && p < tc[g].npairs);
            }
            if (!__ballot(any)) break;
            unsigned info[kGroups];
            float4 pv[kGroups][4], gv[kGroups][4];
            float E[kGroups];
            bool upd[kGroups];
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                FRESH_LANE
                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + (on[g] ? p : 0);
                info[g] = p < 4 ? tc[g].packed(p) : (my_meta[2 * pi] & 0xFFFFu);
                upd[g] = on[g] && tc[g].coeff != 0.0;
                const float4* pp = reinterpret_cast<const float4*>(my_p + pi * 64 + 4 * h);
                const float4* gp = reinterpret_cast<const float4*>(gram + (upd[g] ? (long long)tc[g].sel_g * a.gram_stride + (int)(info[g] & 511u) * 64 : 0) + 4 * h);
#pragma unroll
                for (int t = 0; t < 4; ++t) { pv[g][t] = pp[4 * t]; gv[g][t] = gp[4 * t]; }
                E[g] = my_e[pi];
            }
            static_for<kGroups>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                COUNT(17, 1) COUNT(18, __popcll(__ballot(on[g]) & 0xFFFFull))
                if (on[g]) pair_update(gc, std::true_type{}, p, upd[g], info[g], pv[g], gv[g], E[g]);
            });
        }
        STAMP(2)


        // ---- (9) residual update r -= (q * quant) * row: Vector::Scale then Vector::Subtract, two roundings.  Behind it, what
        //      the next step will read from far away is pulled towards the cache by LDS-DMA into a landing zone nobody reads (no
        //      register, no wait): the Gram rows of the pairs, the filter tiles of a new pair.
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0) return;
            FRESH_LANE
            TileChannel& s = tc[g];
            const bool upd = s.live && s.coeff != 0.0;
            if (upd) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const T scaled = (T)s.coeff * xrow[g][i];
                    r[g][i] = r[g][i] - scaled;
                }
            }
            if (a.out.energy && __ballot(s.live && ended[g])) {    // diagnostic: sum of squares of the final residual, j ascending
                T sq[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) sq[i] = r[g][i] * r[g][i];
                const T e2 = chain_sum(sq, h);
                if (s.live && ended[g] && h == 0) a.out.energy[rec_of(s.t)] = (double)e2;
            }
            unsigned long long pend = __ballot(s.live && !ended[g] && s.fresh >= 0 && s.fresh < 4 && h == 0);
            while (pend) {                                          // a new pair's 16 KiB of filter tiles: 128 lines, two per lane
                const int src = __builtin_ctzll(pend);
                pend &= pend - 1;
                const int blk = __builtin_amdgcn_readlane((int)(s.packed(s.fresh >= 0 && s.fresh < 4 ? s.fresh : 0) & 511u), src);
                const uint16_t* tiles = block_tiles + (long long)blk * (kBlockFilterTiles * 2048) + 64 * lane;
                __builtin_amdgcn_global_load_lds(tiles, my_touch, 4, 0, 0);
                __builtin_amdgcn_global_load_lds(tiles + 4096, my_touch, 4, 0, 0);
            }
            if (ended[g]) s.live = false;
            s.step += 1;
            if (!__ballot(s.live)) unit[g] = -1;
        });
        STAMP(11)
    }
    FRESH_LANE
    if (lane == 0 && a.stats) {
        atomicAdd(a.stats + 0, (unsigned long long)n_mfma);
        atomicAdd(a.stats + 1, (unsigned long long)n_steps);
    }
#ifdef MPC_STAMPS
    if (lane == 0 && a.debug)
        for (int i = 0; i < 24; ++i) atomicAdd(a.debug + i, stamp_acc[i]);
    if (lane == 0 && a.debug && blockIdx.x < 1024) {                           // [24 + 2 b] = first wave in, [25 + 2 b] = last wave out
        atomicMin(a.debug + 24 + 2 * blockIdx.x, wg_begin);
        atomicMax(a.debug + 25 + 2 * blockIdx.x, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
#endif
}

// --------------------------------------------------------------------------------------------------
// Gram table of one channel: G[sel][blk * 64 + row] = float(<row `sel` of the dictionary, row `row` of DetailBasis[blk]>),
// sel = base row (0..509) or 510 + detail row; pad rows and rows left out of the filter copy are 0.  64 x 64 results per
// workgroup, operands staged in LDS, double accumulation.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mp_gram_kernel(const double* __restrict__ base, const double* __restrict__ detail,
                                                      const int32_t* __restrict__ block_rows, const int32_t* __restrict__ block_row_off,
                                                      const uint8_t* __restrict__ shadow, float* __restrict__ gram, int num_base,
                                                      int n_sel, long long stride)
{
    __shared__ double sa[64][65];
    __shared__ double sb[64][65];
    const int blk = blockIdx.x, sel0 = blockIdx.y * 64;
    const int rows = block_rows[blk], off = block_row_off[blk];
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, k = i & 63;
        const int sel = sel0 + rr;
        double va = 0.0;
        if (sel < n_sel) va = sel < num_base ? base[(long long)sel * N + k] : detail[(long long)(sel - num_base) * N + k];
        sa[rr][k] = va;
        double vb = 0.0;
        if (rr < rows && !shadow[off + rr]) vb = detail[(long long)(off + rr) * N + k];
        sb[rr][k] = vb;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k = 0; k < N; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = sa[4 * ty + i][k]; bv[i] = sb[4 * tx + i][k]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(av[i], bv[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sel = sel0 + 4 * ty + i;
        if (sel < n_sel)
            *reinterpret_cast<float4*>(gram + (long long)sel * stride + blk * 64 + 4 * tx) =
                make_float4((float)acc[i][0], (float)acc[i][1], (float)acc[i][2], (float)acc[i][3]);
    }
}

int launch_gram(const double* base, const double* detail, const int32_t* block_rows, const int32_t* block_row_off,
                const uint8_t* shadow, float* gram, int num_base, int n_sel, long long stride, void* stream)
{
    hipLaunchKernelGGL(mp_gram_kernel, dim3((unsigned)num_base, (unsigned)((n_sel + 63) / 64)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), base, detail, block_rows, block_row_off, shadow, gram, num_base, n_sel, stride);
    return (int)hipGetLastError();
}

size_t pursuit_scratch_floats(int workgroups) { return (size_t)workgroups * kWavesMax * kGroups * 16 * kMaxPairs * 64; }
size_t pursuit_scratch_meta(int workgroups) { return (size_t)workgroups * kWavesMax * kGroups * 16 * kMaxPairs * 2; }
size_t pursuit_scratch_bounds(int workgroups) { return (size_t)workgroups * kWavesMax * kGroups * 16 * kMaxPairs; }
int pursuit_units_per_workgroup() { return kWavesDouble * kGroups; }

int launch_pursuit(const PursuitArgs& args, void* stream)
{
    const int workgroups = args.workgroups;
    if (workgroups < 1) return (int)hipErrorInvalidValue;
    if (args.fast)
        hipLaunchKernelGGL((mp_pursuit_kernel<float, kWavesFloat>), dim3((unsigned)workgroups), dim3(64 * kWavesFloat), 0, static_cast<hipStream_t>(stream), args);
    else
        hipLaunchKernelGGL((mp_pursuit_kernel<double, kWavesDouble>), dim3((unsigned)workgroups), dim3(64 * kWavesDouble), 0, static_cast<hipStream_t>(stream), args);
    return (int)hipGetLastError();
}

}  // namespace mpc
