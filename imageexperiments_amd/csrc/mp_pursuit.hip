// mp_pursuit.hip -- product: the PERSISTENT matching-pursuit kernel of the MI355X (gfx950, wave64) tile encoder.
//
// One launch per channel runs the whole K-step pursuit (CalcMPDynamic, MatchingPursuit.cpp:39-74) of every tile-channel of
// that channel; nothing of a tile-channel's state leaves the chip between steps:
//   * a workgroup (8 waves, one per CU) keeps the filter copy of the 510 base rows and of DetailBasis[0] of its channel in
//     LDS (36 tiles of 16 rows x 64 pixels, split-bfloat16 in MFMA operand order, 144 KiB) for its whole life;
//   * a wave owns kGroups column groups of 16 tile-channels; lane (slot = l & 15, h = l >> 4) holds pixels 16h .. 16h+15 of the
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
constexpr int kGroups = 2;                       // column groups (16 tile-channels each) per wave
constexpr int kWaves = 8;                        // waves per workgroup (two per SIMD)
constexpr int kTilesLds = kBaseFilterTiles + kBlockFilterTiles;     // 36
constexpr double W_R = 0.299, W_G = 0.587, W_B = 0.114;            // ImageHelper/inc/misc.h:7-11
constexpr double U_SCALE = 0.436 / (1.0 - 0.114);
constexpr double V_SCALE = 0.615 / (1.0 - 0.299);
constexpr float kSlack = 0x1p-13f;               // E = kSlack |r~| + kAbs bounds the split-bf16 MFMA error (mp_kernels.hip)
constexpr float kAbs = 0x1p-100f;

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

// The two largest values a lane has seen and the row (code) of the largest; sum poisons on NaN / infinity.
struct TopTwo {
    float m1 = 0.0f, m2 = 0.0f, sum = 0.0f;
    int row = -1;
    __device__ __forceinline__ void see(float a, int r)             // a >= 0 (or NaN)
    {
        row = a > m1 ? r : row;
        m2 = __builtin_amdgcn_fmed3f(m1, m2, a);
        m1 = fmaxf(m1, a);
        sum += a;
    }
    // over the four lanes of a slot: top, runner-up (a tie for the top counts), does this lane own the unique top, NaN/Inf seen
    __device__ __forceinline__ void across(float& top, float& second, bool& mine, bool& odd) const
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
        int o = !(sum <= 3.4028234663852886e38f) ? 1 : 0;
        o |= __shfl_xor(o, 16);
        o |= __shfl_xor(o, 32);
        odd = o != 0;
    }
};

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
    int rec = 0;                 // record index (counts[rec], choices[rec * K + step])
    int prev_id = 0;
    int next_off = 0;            // dictionary index where the next unlocked block starts (510 + rows appended so far)
    int off0 = 0;                // index of DetailBasis[0]'s first row once unlocked
    int npairs = 0;
    int fresh = -1;              // pair created by the last step: its approximations come from the MFMAs, not from G
    int fresh_blk = 0;
    unsigned swept = 0;
    bool has0 = false, live = false;
    double coeff = 0.0;          // pending residual update (0 = none): r -= coeff * row
    int sel_g = 0;               // Gram row of the chosen atom (base row, or 510 + detail row)
    const double* sel_row = nullptr;
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

// Sum of the 64 terms held 16 per lane by the four lanes (h = 0..3) of a slot, in pixel order 0..63, starting from 0.0:
// lane h = 0 adds its 16, hands the running sum to h = 1, and so on; the result is broadcast to the four lanes.
__device__ __forceinline__ double chain_sum(const double (&term)[16], int lane)
{
    const int h = lane >> 4;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += term[i];
#pragma unroll
    for (int phase = 1; phase < 4; ++phase) {
        double u = __shfl_up(t, 16);
#pragma unroll
        for (int i = 0; i < 16; ++i) u += term[i];
        t = h >= phase ? u : t;
    }
    return __shfl(t, (lane & 15) + 48);
}

__device__ __forceinline__ void keep_better(double& v, int& i, int& sel, const double*& ptr, double ov, int oi, int osel, const double* optr)
{
    const double a = __builtin_fabs(ov), b = __builtin_fabs(v);
    if (oi >= 0 && (a > b || (a == b && i >= 0 && oi < i))) {       // start: v = 0.0, i = -1: a strict '>' 0 like Select()
        v = ov; i = oi; sel = osel; ptr = optr;
    }
}

}  // namespace

// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWaves, 2) void mp_pursuit_kernel(const PursuitArgs a)
{
    __shared__ uint4 s_tiles[kTilesLds * 256];                      // [tile][operand q][lane], 4 KiB per tile
    {
        const uint4* src_base = reinterpret_cast<const uint4*>(a.base_tiles);
        const uint4* src_blk0 = reinterpret_cast<const uint4*>(a.block_tiles);      // block 0 = the first four tiles
        for (int i = threadIdx.x; i < kTilesLds * 256; i += 64 * kWaves)
            s_tiles[i] = i < kBaseFilterTiles * 256 ? src_base[i] : src_blk0[i - kBaseFilterTiles * 256];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = lane & 15, h = lane >> 4;
    const long long wave_global = (long long)blockIdx.x * kWaves + wave;
    float* const my_p = a.pair_p + wave_global * (kGroups * 16 * kMaxPairs * 64);
    unsigned* const my_meta = a.pair_meta + wave_global * (kGroups * 16 * kMaxPairs * 2);
    float* const my_e = a.pair_e + wave_global * (kGroups * 16 * kMaxPairs);
    const int K = a.K, ch = a.channel;

    int unit[kGroups], step[kGroups];
    double r[kGroups][16];
    TileChannel tc[kGroups];
#pragma unroll
    for (int g = 0; g < kGroups; ++g) { unit[g] = -1; step[g] = 0; }
    bool queue_empty = false;

    for (;;) {
        // ---- (1) refill empty groups from the queue ---------------------------------------------------------------
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] >= 0 || queue_empty) return;
            int u = 0;
            if (lane == 0) u = (int)atomicAdd(a.queue, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= a.n_units) { queue_empty = true; return; }
            unit[g] = u;
            step[g] = 0;
            const long long t = (long long)u * 16 + slot;
            TileChannel s;
            s.live = t < a.n_tc;
            s.next_off = a.num_base;
            if (a.vec_in) {
                s.rec = (int)t;
                if (s.live) load16(r[g], a.vec_in + t * N + 16 * h);
                else
#pragma unroll
                    for (int i = 0; i < 16; ++i) r[g][i] = 0.0;
            } else {
                s.rec = (int)(t * 3 + ch);
                const int tiles_per_frame = a.tiles_x * a.tile_rows;
                const int frame = (int)(t / tiles_per_frame);
                const int tile = (int)(t - (long long)frame * tiles_per_frame);
                const int tx = tile / a.tile_rows, ty = a.tile_row_begin + (tile - tx * a.tile_rows);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int px = tx * 8 + (i & 7), py = ty * 8 + 2 * h + (i >> 3);       // block index dx + 8 dy = 16h + i
                    double v = 0.0;                                                      // zero fill (CompressedImage.cpp:548-552)
                    if (s.live && px < a.width && py < a.height) {
                        const uint8_t* p = a.rgb + (long long)frame * a.frame_stride + (long long)py * a.row_stride + 3 * px;
                        const double red = (double)p[0], green = (double)p[1], blue = (double)p[2];
                        const double Y = (W_R * red + W_G * green + W_B * blue);         // misc.cpp:12-21, same expression order
                        v = ch == 0 ? Y : (ch == 1 ? (U_SCALE * (blue - Y)) : (V_SCALE * (red - Y)));
                    }
                    r[g][i] = v;
                }
            }
            tc[g] = s;
        });
        bool any_unit = false;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any_unit = any_unit || unit[g] >= 0;
        if (!any_unit) break;

        // ---- (2) B operands from the residuals; error bound of this step's MFMA approximations ----------------------
        bf16x8 bh[kGroups][2], bl[kGroups][2];
        float Eb[kGroups], rnorm[kGroups];
        bool nz[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            Eb[g] = 0.0f; rnorm[g] = 0.0f; nz[g] = false;
            if (unit[g] < 0) return;
            double ss = 0.0;
            bool nzl = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const double d = r[g][i];
                nzl = nzl || (d != 0.0);
                const float x = (float)d;
                ss += (double)x * (double)x;
                const unsigned short hb = bf16_of(x);
                const unsigned short lb = bf16_of(x - __uint_as_float((unsigned)hb << 16));
                bh[g][i >> 3][i & 7] = (short)hb;
                bl[g][i >> 3][i & 7] = (short)lb;
            }
            ss += __shfl_xor(ss, 16);
            ss += __shfl_xor(ss, 32);
            int nzi = nzl ? 1 : 0;
            nzi |= __shfl_xor(nzi, 16);
            nzi |= __shfl_xor(nzi, 32);
            nz[g] = nzi != 0;
            rnorm[g] = (float)__builtin_sqrt(ss) * 1.0000002f;
            Eb[g] = kSlack * rnorm[g] + kAbs;
        });

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

        // ---- (2b) a pair created by the last step: first approximations of its 64 rows from the block's filter tiles ---
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0) return;
            unsigned long long pend = __ballot(tc[g].live && tc[g].fresh >= 0 && h == 0);
            float pnew[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) pnew[i] = 0.0f;
            while (pend) {
                const int src = __builtin_ctzll(pend);
                pend &= pend - 1;
                const int blk = __builtin_amdgcn_readlane(tc[g].fresh_blk, src);
                const uint4* tiles = reinterpret_cast<const uint4*>(a.block_tiles) + (long long)blk * (kBlockFilterTiles * 256) + lane;
#pragma unroll
                for (int t = 0; t < kBlockFilterTiles; ++t) {
                    uint4 av[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) av[q] = tiles[t * 256 + q * 64];
                    const f32x4 acc = tile_mfma(av, bh[g], bl[g]);
                    if (slot == src) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) pnew[4 * t + v] = acc[v];
                    }
                }
            }
            if (tc[g].live && tc[g].fresh >= 0) {
                float4* dst = reinterpret_cast<float4*>(my_p + ((long long)(g * 16 + slot) * kMaxPairs + tc[g].fresh) * 64 + 4 * h);
#pragma unroll
                for (int t = 0; t < 4; ++t) dst[4 * t] = make_float4(pnew[4 * t], pnew[4 * t + 1], pnew[4 * t + 2], pnew[4 * t + 3]);
                if (h == 0) my_e[(g * 16 + slot) * kMaxPairs + tc[g].fresh] = Eb[g];
            }
        });

        // ---- (3) pass 1 over the LDS tiles: top two approximations per lane, base rows and DetailBasis[0] apart ----------
        TopTwo tb[kGroups], td[kGroups];
        bool want0[kGroups];
#pragma unroll
        for (int g = 0; g < kGroups; ++g) want0[g] = unit[g] >= 0 && __ballot(tc[g].live && tc[g].has0) != 0;
        bool any0 = false;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any0 = any0 || want0[g];
        {
            const int last = any0 ? kTilesLds : kBaseFilterTiles;
            uint4 cur[4], nxt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = s_tiles[q * 64 + lane];
            for (int t = 0; t < last; ++t) {
                const int tn = t + 1 < last ? t + 1 : t;
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = s_tiles[tn * 256 + q * 64 + lane];
                const bool is_base = t < kBaseFilterTiles;
                const int row0 = (is_base ? t : t - kBaseFilterTiles) * 16 + 4 * h;
                static_for<kGroups>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if (unit[g] < 0 || (!is_base && !want0[g])) return;
                    const f32x4 acc = tile_mfma(cur, bh[g], bl[g]);
                    if (is_base) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) tb[g].see(fabsf(acc[v]), row0 + v);
                    } else {
#pragma unroll
                        for (int v = 0; v < 4; ++v) td[g].see(fabsf(acc[v]), row0 + v);
                    }
                });
#pragma unroll
                for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
            }
        }

        // ---- (4) pairs: Gram update of the kept approximations, top two of their upper bounds ----------------------------
        TopTwo tp[kGroups];
        float lbmax[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            lbmax[g] = -3.0e38f;
            if (unit[g] < 0) return;
            const float c32 = (float)tc[g].coeff;
            const float grow = 0x1p-21f * (fabsf(c32) + rnorm[g]);
            for (int p = 0; __ballot(tc[g].live && p < tc[g].npairs) != 0; ++p) {
                if (!(tc[g].live && p < tc[g].npairs)) continue;
                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + p;
                const unsigned meta = my_meta[2 * pi];
                const int blk = (int)(meta & 511u), rows = (int)((meta >> 9) & 127u);
                float4* pp = reinterpret_cast<float4*>(my_p + pi * 64 + 4 * h);
                float4 pv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) pv[t] = pp[4 * t];
                float E = my_e[pi];
                if (p != tc[g].fresh && tc[g].coeff != 0.0) {
                    const float4* gp = reinterpret_cast<const float4*>(a.gram + (long long)tc[g].sel_g * a.gram_stride + blk * 64 + 4 * h);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 gv = gp[4 * t];
                        pv[t].x = fmaf(-c32, gv.x, pv[t].x);
                        pv[t].y = fmaf(-c32, gv.y, pv[t].y);
                        pv[t].z = fmaf(-c32, gv.z, pv[t].z);
                        pv[t].w = fmaf(-c32, gv.w, pv[t].w);
                        pp[4 * t] = pv[t];
                    }
                    E = (E + grow) * 1.000001f;
                    if (h == 0) my_e[pi] = E;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float vals[4] = {pv[t].x, pv[t].y, pv[t].z, pv[t].w};
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * t + 4 * h + v;
                        const float m = fabsf(vals[v]);
                        const bool real = row < rows;
                        tp[g].see(real ? m + E : 0.0f, (p << 11) | row);
                        lbmax[g] = fmaxf(lbmax[g], real ? m - E : -3.0e38f);
                    }
                }
            }
            lbmax[g] = fmaxf(lbmax[g], __shfl_xor(lbmax[g], 16));
            lbmax[g] = fmaxf(lbmax[g], __shfl_xor(lbmax[g], 32));
        });

        // ---- (5) thresholds; where the runner-up stays below, the single survivor of a segment is known already ----------
        // Every row's exact |p| lies in [|a| - E, |a| + E] (E = Eb for MFMA rows, the pair's own bound otherwise).  With
        // T = the largest lower bound, the row Select() returns -- and every row tying with it -- has |a| + E >= T.
        Survivors sv[kGroups];
        float thr_b[kGroups], T[kGroups];
        bool pass2[kGroups], rescan[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            pass2[g] = false; rescan[g] = false; thr_b[g] = 0.0f; T[g] = 0.0f;
            if (unit[g] < 0) return;
            float top_b, second_b, top_d, second_d, top_p, second_p;
            bool mine_b, mine_d, mine_p, odd_b, odd_d, odd_p;
            tb[g].across(top_b, second_b, mine_b, odd_b);
            td[g].across(top_d, second_d, mine_d, odd_d);
            tp[g].across(top_p, second_p, mine_p, odd_p);
            const bool has0 = tc[g].has0, hasp = tc[g].npairs > 0;
            float t = top_b - Eb[g];
            if (has0) t = fmaxf(t, top_d - Eb[g]);
            if (hasp) t = fmaxf(t, lbmax[g]);
            T[g] = t;
            thr_b[g] = t - Eb[g];
            const bool live = tc[g].live && nz[g];                 // an all-zero residual projects to 0 everywhere: index -1
            const bool unclear_b = !(second_b < thr_b[g]) || (has0 && !(second_d < thr_b[g])) || !(thr_b[g] > 0.0f) || odd_b ||
                                   (has0 && odd_d) || (hasp && odd_p);
            const bool unclear_p = hasp && (!(second_p < t) || odd_p || !(t > 0.0f) || odd_b || (has0 && odd_d));
            pass2[g] = __ballot(live && unclear_b) != 0;
            rescan[g] = live && unclear_p;
            if (live && !pass2[g]) {
                if (mine_b && !(top_b < thr_b[g])) sv[g].push(tb[g].row);
                if (has0 && mine_d && !(top_d < thr_b[g])) sv[g].push((1 << 16) | td[g].row);
            }
            if (live && hasp && !unclear_p && mine_p && !(top_p < t)) sv[g].push((2 << 16) | tp[g].row);
        });

        // ---- (6) pass 2 for groups with an unclear tile-channel: the same MFMAs again, every row at or above the threshold
        bool any2 = false;
#pragma unroll
        for (int g = 0; g < kGroups; ++g) any2 = any2 || pass2[g];
        if (any2) {
            uint4 cur[4], nxt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) cur[q] = s_tiles[q * 64 + lane];
            const int last = any0 ? kTilesLds : kBaseFilterTiles;
            for (int t = 0; t < last; ++t) {
                const int tn = t + 1 < last ? t + 1 : t;
#pragma unroll
                for (int q = 0; q < 4; ++q) nxt[q] = s_tiles[tn * 256 + q * 64 + lane];
                const bool is_base = t < kBaseFilterTiles;
                const int row0 = (is_base ? t : t - kBaseFilterTiles) * 16 + 4 * h;
                const int limit = is_base ? a.num_base : a.rows0;
                static_for<kGroups>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    if (!pass2[g] || (!is_base && !want0[g])) return;
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
        // pairs of an unclear tile-channel: every row whose upper bound reaches T
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0 || !__ballot(rescan[g])) return;
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
                        if (row < rows && !(fabsf(vals[v]) + E < T[g])) sv[g].push((2 << 16) | (p << 11) | row);
                    }
                }
            }
        });

        // ---- (7) exact evaluation of the survivors, one per tile-channel and round, by the four lanes of its slot ----------
        double best_val[kGroups];
        int best_idx[kGroups], best_sel[kGroups];
        const double* best_ptr[kGroups];
        int c0[kGroups], c1[kGroups], c2[kGroups], total[kGroups];
        bool exhaustive[kGroups];
#pragma unroll
        for (int g = 0; g < kGroups; ++g) {
            best_val[g] = 0.0; best_idx[g] = -1; best_sel[g] = 0; best_ptr[g] = a.base;
            c0[g] = __shfl(sv[g].count, slot);
            c1[g] = __shfl(sv[g].count, slot + 16);
            c2[g] = __shfl(sv[g].count, slot + 32);
            total[g] = c0[g] + c1[g] + c2[g] + __shfl(sv[g].count, slot + 48);
            int o = sv[g].overflow ? 1 : 0;
            o |= __shfl_xor(o, 16);
            o |= __shfl_xor(o, 32);
            exhaustive[g] = o != 0;
        }
        // what a code means for this lane's tile-channel: the row's address, its dictionary index and its Gram row
        auto resolve = [&](auto gc, int code, const double*& ptr, int& idx, int& sel) {
            constexpr int g = decltype(gc)::value;
            const int seg = code >> 16, row = code & 2047;
            if (seg == 0) { ptr = a.base + (long long)row * N; idx = row; sel = row; }
            else if (seg == 1) { ptr = a.detail + (long long)row * N; idx = tc[g].off0 + row; sel = a.num_base + row; }
            else {
                const int p = (code >> 11) & 31;
                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + p;
                const unsigned meta = my_meta[2 * pi];
                const int drow = (int)my_meta[2 * pi + 1] + row;
                ptr = a.detail + (long long)drow * N;
                idx = (int)(meta >> 16) + row;
                sel = a.num_base + drow;
            }
        };
        auto evaluate = [&](auto gc, bool on, int code) {
            constexpr int g = decltype(gc)::value;
            const double* ptr = a.base;
            int idx = -1, sel = 0;
            if (on) resolve(gc, code, ptr, idx, sel);
            double x[16];
            load16(x, ptr + 16 * h);
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = x[i] * r[g][i];             // the reference's l * r, rounded to double
            const double p = chain_sum(x, lane);
            if (on) keep_better(best_val[g], best_idx[g], best_sel[g], best_ptr[g], p, idx, sel, ptr);
        };
        for (int k = 0;; ++k) {
            bool on[kGroups];
            bool any = false;
#pragma unroll
            for (int g = 0; g < kGroups; ++g) {
                on[g] = unit[g] >= 0 && tc[g].live && !exhaustive[g] && k < total[g];
                any = any || on[g];
            }
            if (!__ballot(any)) break;
            static_for<kGroups>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (!__ballot(on[g])) return;
                const int owner = k < c0[g] ? 0 : (k < c0[g] + c1[g] ? 1 : (k < c0[g] + c1[g] + c2[g] ? 2 : 3));
                const int local = k - (owner == 0 ? 0 : (owner == 1 ? c0[g] : (owner == 2 ? c0[g] + c1[g] : c0[g] + c1[g] + c2[g])));
                const int code = __shfl(sv[g].at(local & 3), slot + 16 * owner);
                evaluate(gc, on[g], code);
            });
        }
        // a tile-channel whose survivors did not fit (many exact ties, NaN): every row it can choose from, in order
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0) return;
            const bool ex = tc[g].live && nz[g] && exhaustive[g];
            if (!__ballot(ex)) return;
            const int n_all = a.num_base + 64 + 64 * tc[g].npairs;
            for (int n = 0; __ballot(ex && n < n_all) != 0; ++n) {
                bool on = ex && n < n_all;
                int code = 0;
                if (n < a.num_base) code = n;
                else if (n < a.num_base + 64) { code = (1 << 16) | (n - a.num_base); on = on && tc[g].has0 && n - a.num_base < a.rows0; }
                else {
                    const int p = (n - a.num_base - 64) >> 6, row = (n - a.num_base - 64) & 63;
                    code = (2 << 16) | (p << 11) | row;
                    if (on) on = row < (int)((my_meta[2 * ((long long)(g * 16 + slot) * kMaxPairs + p)] >> 9) & 127u);
                }
                if (!__ballot(on)) continue;
                evaluate(gc, on, code);
            }
        });

        // ---- (8) finish: delta / zig-zag, quantise, record, unlock, termination (MatchingPursuit.cpp:50-71) ------------------
        bool ended[kGroups];
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            ended[g] = false;
            if (unit[g] < 0) return;
            TileChannel& s = tc[g];
            s.coeff = 0.0;
            s.fresh = -1;
            if (!s.live) return;
            s.swept += (unsigned)s.next_off;                       // rows of this step's dictionary (SURVEY 8d "S")
            int count = 0;
            unsigned record = 0;
            if (best_idx[g] < 0) {
                ended[g] = true; count = step[g];
            } else {
                const int id = best_idx[g];
                const unsigned delta = step[g] > 0 ? (((unsigned)(id - s.prev_id) << 1) ^ (unsigned)((id - s.prev_id) >> 31)) : (unsigned)id;
                s.prev_id = id;
                const double qstep = a.quant[step[g]];
                const int q = (int)__builtin_round(best_val[g] / qstep);
                const unsigned zz = ((unsigned)q << 1) ^ (unsigned)(q >> 31);
                record = (delta & 0xFFFFu) | ((zz & 0xFFFFu) << 16);
                if (q == 0) {
                    ended[g] = true; count = step[g];
                } else {
                    s.coeff = qstep * (double)q;
                    s.sel_g = best_sel[g];
                    s.sel_row = best_ptr[g];
                    if (step[g] + 1 == K) { ended[g] = true; count = K; }
                    else if (id < a.num_base) {                     // unlock DetailBasis[id]; a repeat is indexed, not swept again
                        const int rows = a.block_rows[id];
                        bool repeat = id == 0 && s.has0;
                        for (int p = 0; p < s.npairs; ++p)
                            repeat = repeat || (int)(my_meta[2 * ((long long)(g * 16 + slot) * kMaxPairs + p)] & 511u) == id;
                        if (!repeat) {
                            if (id == 0) { s.has0 = true; s.off0 = s.next_off; }
                            else {
                                const long long pi = (long long)(g * 16 + slot) * kMaxPairs + s.npairs;
                                if (h == 0) {
                                    my_meta[2 * pi] = (unsigned)id | ((unsigned)rows << 9) | ((unsigned)s.next_off << 16);
                                    my_meta[2 * pi + 1] = (unsigned)a.block_row_off[id];
                                }
                                s.fresh = s.npairs;
                                s.fresh_blk = id;
                                s.npairs += 1;
                            }
                        }
                        s.next_off += rows;
                    }
                }
            }
            if (h == 0) {
                a.out.choices[(long long)s.rec * K + step[g]] = record;
                if (ended[g]) {
                    a.out.counts[s.rec] = (uint16_t)count;
                    if (a.out.swept) a.out.swept[s.rec] = s.swept;
                }
            }
        });
        // the lanes of a slot wrote the pair's meta words from h == 0 only: make them visible to the other three before use
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- (9) residual update r -= (q * quant) * row: Vector::Scale then Vector::Subtract, two roundings --------------------
        static_for<kGroups>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (unit[g] < 0) return;
            TileChannel& s = tc[g];
            if (__ballot(s.live && s.coeff != 0.0)) {
                double x[16];
                load16(x, (s.live && s.coeff != 0.0 ? s.sel_row : a.base) + 16 * h);
                if (s.live && s.coeff != 0.0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const double scaled = s.coeff * x[i];
                        r[g][i] = r[g][i] - scaled;
                    }
                }
            }
            if (a.out.energy && __ballot(s.live && ended[g])) {    // diagnostic: sum of squares of the final residual, j ascending
                double sq[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) sq[i] = r[g][i] * r[g][i];
                const double e2 = chain_sum(sq, lane);
                if (s.live && ended[g] && h == 0) a.out.energy[s.rec] = e2;
            }
            if (ended[g]) s.live = false;
            step[g] += 1;
            if (!__ballot(s.live)) unit[g] = -1;
        });
    }
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

size_t pursuit_scratch_floats(int workgroups) { return (size_t)workgroups * kWaves * kGroups * 16 * kMaxPairs * 64; }
size_t pursuit_scratch_meta(int workgroups) { return (size_t)workgroups * kWaves * kGroups * 16 * kMaxPairs * 2; }
size_t pursuit_scratch_bounds(int workgroups) { return (size_t)workgroups * kWaves * kGroups * 16 * kMaxPairs; }
int pursuit_units_per_workgroup() { return kWaves * kGroups; }

int launch_pursuit(const PursuitArgs& args, int workgroups, void* stream)
{
    if (workgroups < 1) workgroups = 1;
    hipLaunchKernelGGL(mp_pursuit_kernel, dim3((unsigned)workgroups), dim3(64 * kWaves), 0, static_cast<hipStream_t>(stream), args);
    return (int)hipGetLastError();
}

}  // namespace mpc
