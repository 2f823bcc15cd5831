/*
 * mpo_mp.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of
 *   createQuantizationTables   CompressionLib/src/CompressedImage.cpp:16-166
 *   YUVFromRGB / RGBFromYUV     ImageHelper/src/misc.cpp:7-36, inc/misc.h:7-11
 *   Select / CalcMPDynamic      CompressionLib/src/MatchingPursuit.cpp:7-74
 *   dynamicBasis                CompressionLib/src/CompressedImage.cpp:212-250
 *   FromCoeffsDynamic           CompressionLib/src/MatchingPursuit.cpp:109-128
 *   tile gather of encodeImage  CompressionLib/src/CompressedImage.cpp:535-573
 *   Multiply / Scale / Subtract SimpleMatrix/src/mathmatrix.cpp:426-449,
 *                               mathvector.cpp:116-148
 * The dynamic dictionary is walked through a list of row blocks instead of
 * being re-materialised each step (same rows, same order, same arithmetic;
 * the reference's per-step allocate+memcpy is not reproduced, so this oracle
 * is a *faster* CPU baseline than the reference itself).
 */
#include "mpo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- variance tables: data constants of CompressedImage.cpp:16-122 ---- */
static const double DECAY[3] = {0.902045039488061, 0.896332644824969, 0.897340618787505};
static const double VAR[3][32] = {
    {1449455.61399403, 30867.8722232759, 4879.76236869648, 2065.81004100418,
     1177.78544096912, 754.545827240537, 519.229145237154, 375.509017928094,
     281.02055698585, 216.291896608802, 170.433377219481, 137.390795170594,
     111.760859784514, 92.465892223227, 77.4762763021059, 65.5657474800836,
     55.9694631412915, 48.1296222409122, 41.6533463654379, 36.2103188098409,
     31.620636785048, 27.6994619197188, 24.3350442530727, 21.416843198437,
     18.879120562683, 16.6557974119762, 14.7004164720848, 12.9791457307132,
     11.4562911334688, 10.1096285629269, 8.9223405657796, 7.86882234976579},
    {51995.6231219068, 814.727839313831, 108.677634702502, 49.7911952400269,
     28.7429354781437, 19.0770865041308, 13.7134473152652, 10.4220864760027,
     8.24661056024952, 6.70462984374069, 5.55959248319828, 4.6929026981696,
     4.00644915447918, 3.4569689589025, 3.00626981236079, 2.63088816527246,
     2.31483558804887, 2.04636560444097, 1.81780225821739, 1.61760270635648,
     1.44182998235134, 1.29183839884355, 1.15688638919911, 1.0404065817416,
     0.938042133945239, 0.844584472128933, 0.764573763603021, 0.691885170456949,
     0.629607793503852, 0.573021503073893, 0.52497651127712, 0.483288166342945},
    {60578.6241767756, 617.61939120778, 70.9553277465942, 31.4166652349442,
     16.8206825627114, 10.4578171000126, 7.14592512982323, 5.22663706003167,
     4.01504652955091, 3.20518029980816, 2.64504709802794, 2.2477896896281,
     1.94630992425302, 1.72002308826788, 1.53680040297081, 1.39452712284538,
     1.27393363160348, 1.17553215497528, 1.08841824408173, 1.01505869329656,
     0.950595903582374, 0.893571526043102, 0.841836849604898, 0.792090537556817,
     0.74756288452653, 0.704561007977196, 0.665399335594415, 0.631274499723472,
     0.597331898015673, 0.568852846586831, 0.539371538069597, 0.513182721162335}};

/* CompressedImage.cpp:124-166 */
void mpo_quant_tables(int K, int bs, double bpp, double *qY, double *qU, double *qV)
{
    double allocated = 0.0;
    double alloc[3 * MPO_MAX_K], var[3 * MPO_MAX_K];
    for (int i = 0; i < 3 * K; i++) alloc[i] = 0.0;
    for (int i = 0; i < K; i++) {
        var[i] = VAR[0][i];
        var[i + K] = VAR[1][i];
        var[i + 2 * K] = VAR[2][i];
    }
    while ((allocated / (double)(bs * bs)) < bpp) {
        int index = 0;                              /* std::max_element: first maximum */
        for (int i = 1; i < 3 * K; i++)
            if (var[index] < var[i]) index = i;
        alloc[index] += 1.0;
        var[index] /= 2.0;
        allocated += 1.0;
        double min = 1.0;
        if (index == 0 || index == K || index == 2 * K)
            min = (double)bs;
        if (255.0 * (double)bs * pow(0.5, alloc[index]) < min)
            var[index] = 0.0;
    }
    double *q[3] = {qY, qU, qV};
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) {
            double min = (i == 0) ? (double)bs : 1.0;
            double v = ceil(255.0 * (double)bs * pow(DECAY[ch], (double)i) * pow(0.5, alloc[i + ch * K]));
            q[ch][i] = (v < min) ? min : v;         /* std::max(v, min) */
        }
}

/* misc.cpp:12-21; weights misc.h:7-11 */
void mpo_yuv_from_rgb(uint8_t r8, uint8_t g8, uint8_t b8, double *y, double *u, double *v)
{
    double red = (double)r8, green = (double)g8, blue = (double)b8;
    double Y = (0.299 * red + 0.587 * green + 0.114 * blue);
    *y = Y;
    *u = ((0.436 / (1.0 - 0.114)) * (blue - Y));
    *v = ((0.615 / (1.0 - 0.299)) * (red - Y));
}

static uint8_t clamp_round_u8(double x)
{
    double r = round(x);
    if (r < 0.0) r = 0.0;
    if (r > 255.0) r = 255.0;       /* std::clamp(round(x),0,255); NaN would pass through, never produced */
    return (uint8_t)r;
}

/* misc.cpp:28-36 */
void mpo_rgb_from_yuv(double y, double u, double v, uint8_t *r, uint8_t *g, uint8_t *b)
{
    *r = clamp_round_u8(y + 1.13983 * v);
    *g = clamp_round_u8(y - 0.39466 * u - 0.58060 * v);
    *b = clamp_round_u8(y + 2.03211 * u);
}

/* one GEMV row, mathmatrix.cpp:436-444: tot starts at 0.0, tot += l*r in order */
static inline double row_dot(const double *row, const double *r, int N)
{
    double tot = 0.0;
    for (int j = 0; j < N; j++)
        tot += row[j] * r[j];
    return tot;
}

/* MatchingPursuit.cpp:39-74 with Select :7-25 inlined */
int mpo_calc_mp(const mpo_ctx *c, int ch, const double *quant, const double *in,
                uint16_t *delta_id, uint16_t *int_coeff, double *resid, uint32_t *swept)
{
    const int N = c->N, K = c->K;
    double r[64];
    memcpy(r, in, sizeof(double) * (size_t)N);
    uint32_t S = 0;
    int prev = 0, count = K;
    for (int reps = 0; reps < K; reps++) {
        /* dictionary(reps, results): base rows, then DetailBasis[choice_i] for every
         * earlier step whose decoded choice is a base atom, duplicates included
         * (CompressedImage.cpp:219-248) */
        int index = -1, id = 0;
        double best = 0.0;
        const double *best_row = NULL;
        for (int a = 0; a < c->nbase; a++, id++) {
            const double *row = c->base + (size_t)a * N;
            double p = row_dot(row, r, N);
            if (fabs(p) > fabs(best)) { best = p; index = id; best_row = row; }
        }
        int choice = 0;
        for (int i = 0; i < reps; i++) {
            if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
            else choice = (int)delta_id[0];
            if (choice >= 0 && choice < c->nbase) {
                const double *blk = c->det[ch] + c->det_off[choice] * N;
                for (int a = 0; a < c->det_rows[choice]; a++, id++) {
                    const double *row = blk + (size_t)a * N;
                    double p = row_dot(row, r, N);
                    if (fabs(p) > fabs(best)) { best = p; index = id; best_row = row; }
                }
            }
        }
        S += (uint32_t)id;
        if (index < 0) {                                   /* :50-54 */
            delta_id[reps] = 0;
            int_coeff[reps] = 0;
            count = reps;
            break;
        }
        if (reps > 0)
            delta_id[reps] = (uint16_t)mpo_zigzag_enc(index - prev);
        else
            delta_id[reps] = (uint16_t)index;
        prev = index;
        int q = (int)round(best / quant[reps]);
        int_coeff[reps] = (uint16_t)mpo_zigzag_enc(q);
        double coeff = quant[reps] * (double)q;
        if (q == 0) {                                      /* :66-69 */
            count = reps;
            break;
        }
        for (int j = 0; j < N; j++) {
            double scaled = coeff * best_row[j];           /* Vector::Scale    mathvector.cpp:140 */
            r[j] = r[j] - scaled;                          /* Vector::Subtract mathvector.cpp:116 */
        }
    }
    if (resid) memcpy(resid, r, sizeof(double) * (size_t)N);
    if (swept) *swept = S;
    return count;
}

/* MatchingPursuit.cpp:109-128 */
void mpo_from_coeffs(const mpo_ctx *c, int ch, const double *quant, int count,
                     const uint16_t *delta_id, const uint16_t *int_coeff, double *out)
{
    const int N = c->N;
    /* dictionary(K=count, coeffs): row table instead of a copy */
    const double *rows[64 * (MPO_MAX_K + 1) + 512];
    int nrows = 0;
    for (int a = 0; a < c->nbase; a++) rows[nrows++] = c->base + (size_t)a * N;
    int choice = 0;
    for (int i = 0; i < count; i++) {
        if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
        else choice = (int)delta_id[0];
        if (choice >= 0 && choice < c->nbase) {
            const double *blk = c->det[ch] + c->det_off[choice] * N;
            for (int a = 0; a < c->det_rows[choice]; a++) rows[nrows++] = blk + (size_t)a * N;
        }
    }
    for (int j = 0; j < N; j++) out[j] = 0.0;
    choice = 0;
    for (int i = 0; i < count; i++) {
        if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
        else choice = (int)delta_id[0];
        double coeff = quant[i] * (double)mpo_zigzag_dec((uint32_t)int_coeff[i]);
        if (choice < 0 || choice >= nrows) continue;       /* reference would throw (bounds-checked []) */
        const double *basis = rows[choice];
        for (int j = 0; j < N; j++)
            out[j] += basis[j] * coeff;
    }
}

/* CompressedImage.cpp:538-554; image<rgb> is row-major, 3 B/px (image.h:123-131) */
void mpo_gather_tile(const uint8_t *rgb, int W, int H, int bs, int x0, int y0, double *yuv)
{
    const int N = bs * bs;
    for (int dx = 0; dx < bs; dx++) {
        int u = x0 + dx;
        for (int dy = 0; dy < bs; dy++) {
            int v = y0 + dy;
            double Y = 0.0, U = 0.0, V = 0.0;
            if (u < W && v < H) {
                const uint8_t *p = rgb + 3 * ((size_t)v * W + u);
                mpo_yuv_from_rgb(p[0], p[1], p[2], &Y, &U, &V);
            }
            yuv[dx + bs * dy] = Y;
            yuv[N + dx + bs * dy] = U;
            yuv[2 * N + dx + bs * dy] = V;
        }
    }
}

void mpo_encode_tiles(const mpo_ctx *c, const uint8_t *rgb, int W, int H,
                      const double *qY, const double *qU, const double *qV,
                      int tx_begin, int tx_end,
                      uint16_t *counts, uint16_t *delta, uint16_t *coef,
                      double *energy, uint32_t *swept)
{
    const int bs = c->bs, N = c->N, K = c->K;
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    const double *q[3] = {qY, qU, qV};
    if (tx_end > tilesX) tx_end = tilesX;
    for (int tx = tx_begin; tx < tx_end; tx++)
        for (int ty = 0; ty < tilesY; ty++) {
            size_t t = (size_t)tx * tilesY + ty;
            double yuv[3 * 64], res[64];
            mpo_gather_tile(rgb, W, H, bs, tx * bs, ty * bs, yuv);
            for (int ch = 0; ch < 3; ch++) {
                uint16_t d[MPO_MAX_K], k[MPO_MAX_K];
                uint32_t S;
                memset(d, 0, sizeof d);
                memset(k, 0, sizeof k);
                int cnt = mpo_calc_mp(c, ch, q[ch], yuv + ch * N, d, k, res, &S);
                size_t o = t * 3 + ch;
                if (counts) counts[o] = (uint16_t)cnt;
                if (delta) memcpy(delta + o * K, d, sizeof(uint16_t) * (size_t)K);
                if (coef) memcpy(coef + o * K, k, sizeof(uint16_t) * (size_t)K);
                if (energy) {
                    double e = 0.0;
                    for (int j = 0; j < N; j++) e += res[j] * res[j];
                    energy[o] = e;
                }
                if (swept) swept[o] = S;
            }
        }
}

/* calculatePSNR CompressedImage.cpp:343-357 (x outer, y inner summation order) */
double mpo_psnr(const uint8_t *a, const uint8_t *b, int W, int H)
{
    double se = 0.0;
    for (int x = 0; x < W; x++)
        for (int y = 0; y < H; y++) {
            const uint8_t *pa = a + 3 * ((size_t)y * W + x), *pb = b + 3 * ((size_t)y * W + x);
            double r = (double)pb[0] - (double)pa[0];
            double g = (double)pb[1] - (double)pa[1];
            double bl = (double)pb[2] - (double)pa[2];
            se += r * r + g * g + bl * bl;
        }
    double mse = se / (double)((size_t)W * (size_t)H);
    return 20.0 * log10(3.0 * 255.0) - 10.0 * log10(mse);
}

/* ---- std::mt19937 (standard MT19937, 32-bit) for the BASELINE.md 3 generator ---- */
typedef struct { uint32_t s[624]; int i; } mt_state;

static void mt_seed(mt_state *m, uint32_t seed)
{
    m->s[0] = seed;
    for (int i = 1; i < 624; i++)
        m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
    m->i = 624;
}

static uint32_t mt_next(mt_state *m)
{
    if (m->i >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (m->s[k] & 0x80000000u) | (m->s[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = m->s[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            m->s[k] = v;
        }
        m->i = 0;
    }
    uint32_t y = m->s[m->i++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static uint8_t clamp255(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void mpo_synth_frame(uint8_t *rgb, int W, int H, uint32_t seed)
{
    mt_state m;
    mt_seed(&m, seed);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int n = (int)(mt_next(&m) % 32u) - 16;
            uint8_t *p = rgb + 3 * ((size_t)y * W + x);
            p[0] = clamp255(x * 255 / W + n);
            p[1] = clamp255(y * 255 / H + n);
            p[2] = clamp255(128 + 3 * n);
        }
}

/* ---- "-s" patch statistics: Compression.cpp:200-302, math::Stat SimpleMatrix/src/covariance.cpp:5-25 ---- */
typedef struct { double N, min, max, mean, sumSq; } mpo_stat;
struct mpo_patch_stats {
    mt_state rng;
    int K;
    mpo_stat s[3][2][MPO_MAX_K];      /* [channel][0 = intCoeff, 1 = deltaId][step] */
};

static void stat_update(mpo_stat *s, double val)                   /* covariance.cpp:5-21 (Welford) */
{
    if (s->N == 0) {                       /* first sample defines min, max and mean */
        s->N = 1.0;
        s->min = s->max = s->mean = val;
        s->sumSq = 0.0;
        return;
    }
    if (val < s->min) s->min = val;
    if (val > s->max) s->max = val;
    s->N += 1.0;
    const double before = val - s->mean;   /* distance to the old mean ... */
    s->mean += before / s->N;
    const double after = val - s->mean;    /* ... and to the new one */
    s->sumSq += before * after;
}

mpo_patch_stats *mpo_patch_stats_create(int K, uint32_t seed)      /* Compression.cpp:210-220 */
{
    mpo_patch_stats *p = (mpo_patch_stats *)calloc(1, sizeof *p);
    if (!p) return NULL;
    p->K = K;
    mt_seed(&p->rng, seed);
    return p;
}

void mpo_patch_stats_destroy(mpo_patch_stats *p) { free(p); }

/* one image of the loop at Compression.cpp:226-273; all quantisers 1.0 (:221-225) */
void mpo_patch_stats_add_image(mpo_patch_stats *p, const mpo_ctx *c, const uint8_t *rgb, int W, int H, int patches)
{
    const int bs = c->bs, N = c->N, K = p->K;
    if (W < bs || H < bs) return;                                  /* :233-236 */
    double quant[MPO_MAX_K];
    for (int i = 0; i < K; i++) quant[i] = 1.0;
    double *yuv = (double *)malloc(sizeof(double) * 3 * N);
    uint16_t delta[MPO_MAX_K], coef[MPO_MAX_K];
    for (int reps = 0; reps < patches; reps++) {
        int x = (int)(mt_next(&p->rng) % (uint32_t)(W - bs));      /* :238 */
        int y = (int)(mt_next(&p->rng) % (uint32_t)(H - bs));      /* :239 */
        mpo_gather_tile(rgb, W, H, bs, x, y, yuv);                 /* :240-248: patch[offx + offy*bs], always inside */
        for (int ch = 0; ch < 3; ch++) {                           /* :249-269 */
            int count = mpo_calc_mp(c, ch, quant, yuv + ch * N, delta, coef, NULL, NULL);
            for (int i = 0; i < count; i++) {
                stat_update(&p->s[ch][0][i], (double)coef[i]);
                stat_update(&p->s[ch][1][i], (double)delta[i]);
            }
        }
    }
    free(yuv);
}

/* out[3][2][K][5] = N, min, max, mean, sumSq */
void mpo_patch_stats_read(const mpo_patch_stats *p, double *out)
{
    for (int ch = 0; ch < 3; ch++)
        for (int kind = 0; kind < 2; kind++)
            for (int i = 0; i < p->K; i++) {
                const mpo_stat *s = &p->s[ch][kind][i];
                double *o = out + (((size_t)ch * 2 + kind) * p->K + i) * 5;
                o[0] = s->N; o[1] = s->min; o[2] = s->max; o[3] = s->mean; o[4] = s->sumSq;
            }
}

/* accessor for tests: the variance constants above vs Data/stats.txt */
double mpo_variance_constant(int ch, int i) { return VAR[ch][i]; }
double mpo_decay_constant(int ch) { return DECAY[ch]; }
