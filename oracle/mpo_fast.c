/*
 * mpo_fast.c -- ORACLE (test infrastructure, NOT product code): the `...Fast` (float) flavour of the tile path.
 *
 * PARITY UNPINNED.  The reference's Fast path is Eigen arithmetic (MatchingPursuit.cpp:27-37,76-107,130-147: `dictionary *
 * residual`, `cwiseAbs().maxCoeff`, `residual -= coeff * newEntry`) on a dictionary built in float with Eigen's
 * SelfAdjointEigenSolver<MatrixXf> (BasisSet.cpp:162-185).  Eigen is an empty submodule of the reference tree (branch 3.4,
 * commit unknown), no reference test or fixture holds a Fast result, and the order in which Eigen's GEMV adds its products
 * is not specified.  What is restated here is therefore a DEFINITION of the float mode, chosen to be the reference's
 * statements read literally in float:
 *   dictionary   the double dictionary (bit-identical to the reference's double one) rounded to float, row by row
 *   projection   p_i = sum_j d_ij * r_j in float, j ascending, every product and sum rounded separately
 *   selection    the first maximum of |p| (Eigen's maxCoeff visitor: strict >, so an all-zero residual selects row 0 with
 *                coefficient 0 -- the step then quantises to 0 and ends the pursuit exactly as index -1 does in double)
 *   quantise     (int)round(coeff / quant[reps]) with float division; coeff = quant[reps] * (float)quantised
 *   update       r_j = r_j - coeff * e_j, two roundings
 *   input        YUVFromRGB in double, then static_cast<float> (CompressedImage.cpp:601-605)
 *   decode       results_j += coeff * row_j in float (MatchingPursuit.cpp:130-147), RGBFromYUV on the floats widened to double
 * The product's float kernels are checked bit for bit against THIS, and against the double path by PSNR / size
 * (tests/test_gpu_fast_mode.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "mpo.h"

struct mpo_fast {
    const mpo_ctx *c;
    float *base;            /* [nbase][N]              */
    float *det[3];          /* [det_off[nbase]][N]     */
};

mpo_fast *mpo_fast_create(const mpo_ctx *c)
{
    mpo_fast *f = (mpo_fast *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->c = c;
    const size_t N = (size_t)c->N, nb = (size_t)c->nbase * N, nd = c->det_off[c->nbase] * N;
    f->base = (float *)malloc(sizeof(float) * (nb ? nb : 1));
    for (size_t i = 0; i < nb; i++) f->base[i] = (float)c->base[i];
    for (int ch = 0; ch < 3; ch++) {
        f->det[ch] = (float *)malloc(sizeof(float) * (nd ? nd : 1));
        for (size_t i = 0; i < nd; i++) f->det[ch][i] = (float)c->det[ch][i];
    }
    return f;
}

void mpo_fast_destroy(mpo_fast *f)
{
    if (!f) return;
    free(f->base);
    for (int ch = 0; ch < 3; ch++) free(f->det[ch]);
    free(f);
}

static float row_dot_f(const float *row, const float *r, int N)
{
    float tot = 0.0f;
    for (int j = 0; j < N; j++) {
        float prod = row[j] * r[j];
        tot = tot + prod;
    }
    return tot;
}

/* CalcMPDynamicFast MatchingPursuit.cpp:76-107 with SelectFast :27-37 and dynamicBasisFast CompressedImage.cpp:252-289 */
int mpo_calc_mp_fast(const mpo_fast *f, int ch, const float *quant, const float *in,
                     uint16_t *delta_id, uint16_t *int_coeff, float *resid, uint32_t *swept)
{
    const mpo_ctx *c = f->c;
    const int N = c->N, K = c->K;
    float r[64];
    memcpy(r, in, sizeof(float) * (size_t)N);
    uint32_t S = 0;
    int prev = 0, count = K;
    for (int reps = 0; reps < K; reps++) {
        int index = -1, id = 0;
        float best = 0.0f, best_abs = -1.0f;               /* maxCoeff: the first maximum, also of an all-zero vector */
        const float *best_row = NULL;
        for (int a = 0; a < c->nbase; a++, id++) {
            const float *row = f->base + (size_t)a * N;
            float p = row_dot_f(row, r, N);
            if (fabsf(p) > best_abs) { best = p; best_abs = fabsf(p); index = id; best_row = row; }
        }
        int choice = 0;
        for (int i = 0; i < reps; i++) {
            if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
            else choice = (int)delta_id[0];
            if (choice >= 0 && choice < c->nbase) {
                const float *blk = f->det[ch] + c->det_off[choice] * N;
                for (int a = 0; a < c->det_rows[choice]; a++, id++) {
                    const float *row = blk + (size_t)a * N;
                    float p = row_dot_f(row, r, N);
                    if (fabsf(p) > best_abs) { best = p; best_abs = fabsf(p); index = id; best_row = row; }
                }
            }
        }
        S += (uint32_t)id;
        if (index < 0) {                                   /* :84-88 (a NaN residual: nothing compares greater) */
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
        float ratio = best / quant[reps];
        int q = (int)roundf(ratio);
        int_coeff[reps] = (uint16_t)mpo_zigzag_enc(q);
        float coeff = quant[reps] * (float)q;
        if (q == 0) {                                      /* :100-102 */
            count = reps;
            break;
        }
        for (int j = 0; j < N; j++) {
            float scaled = coeff * best_row[j];
            r[j] = r[j] - scaled;
        }
    }
    if (resid) memcpy(resid, r, sizeof(float) * (size_t)N);
    if (swept) *swept = S;
    return count;
}

/* FromCoeffsDynamicFast MatchingPursuit.cpp:130-147 */
void mpo_from_coeffs_fast(const mpo_fast *f, int ch, const float *quant, int count,
                          const uint16_t *delta_id, const uint16_t *int_coeff, float *out)
{
    const mpo_ctx *c = f->c;
    const int N = c->N;
    const float *rows[64 * (MPO_MAX_K + 1) + 512];
    int nrows = 0;
    for (int a = 0; a < c->nbase; a++) rows[nrows++] = f->base + (size_t)a * N;
    int choice = 0;
    for (int i = 0; i < count; i++) {
        if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
        else choice = (int)delta_id[0];
        if (choice >= 0 && choice < c->nbase) {
            const float *blk = f->det[ch] + c->det_off[choice] * N;
            for (int a = 0; a < c->det_rows[choice]; a++) rows[nrows++] = blk + (size_t)a * N;
        }
    }
    for (int j = 0; j < N; j++) out[j] = 0.0f;
    choice = 0;
    for (int i = 0; i < count; i++) {
        if (i > 0) choice = choice + mpo_zigzag_dec((uint32_t)delta_id[i]);
        else choice = (int)delta_id[0];
        float coeff = quant[i] * (float)mpo_zigzag_dec((uint32_t)int_coeff[i]);
        if (choice < 0 || choice >= nrows) continue;
        const float *basis = rows[choice];
        for (int j = 0; j < N; j++) {
            float term = coeff * basis[j];
            out[j] = out[j] + term;
        }
    }
}

/* encodeImageFast's tile loop CompressedImage.cpp:592-630; outputs as mpo_encode_tiles (energy = sum r^2 in float, j ascending) */
void mpo_encode_tiles_fast(const mpo_fast *f, const uint8_t *rgb, int W, int H,
                           const double *qY, const double *qU, const double *qV,
                           int tx_begin, int tx_end,
                           uint16_t *counts, uint16_t *delta, uint16_t *coef,
                           double *energy, uint32_t *swept)
{
    const mpo_ctx *c = f->c;
    const int bs = c->bs, N = c->N, K = c->K;
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    const double *q64[3] = {qY, qU, qV};
    float q[3][MPO_MAX_K];
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) q[ch][i] = (float)q64[ch][i];
    if (tx_end > tilesX) tx_end = tilesX;
    for (int tx = tx_begin; tx < tx_end; tx++)
        for (int ty = 0; ty < tilesY; ty++) {
            size_t t = (size_t)tx * tilesY + ty;
            double yuv[3 * 64];
            mpo_gather_tile(rgb, W, H, bs, tx * bs, ty * bs, yuv);
            for (int ch = 0; ch < 3; ch++) {
                float in[64], res[64];
                for (int j = 0; j < N; j++) in[j] = (float)yuv[ch * N + j];
                uint16_t d[MPO_MAX_K], k[MPO_MAX_K];
                uint32_t S = 0;
                memset(d, 0, sizeof d);
                memset(k, 0, sizeof k);
                int cnt = mpo_calc_mp_fast(f, ch, q[ch], in, d, k, res, &S);
                size_t o = t * 3 + ch;
                if (counts) counts[o] = (uint16_t)cnt;
                if (delta) memcpy(delta + o * K, d, sizeof(uint16_t) * (size_t)K);
                if (coef) memcpy(coef + o * K, k, sizeof(uint16_t) * (size_t)K);
                if (energy) {
                    float e = 0.0f;
                    for (int j = 0; j < N; j++) {
                        float sq = res[j] * res[j];
                        e = e + sq;
                    }
                    energy[o] = (double)e;
                }
                if (swept) swept[o] = S;
            }
        }
}

/* encodeImageFast CompressedImage.cpp:578-633 (the container is the double path's: writeCompressed) */
uint8_t *mpo_encode_image_fast(const mpo_fast *f, const uint8_t *rgb, int W, int H,
                               const double *qY, const double *qU, const double *qV, size_t *nbytes)
{
    const mpo_ctx *c = f->c;
    const int bs = c->bs, K = c->K;
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    const size_t tiles = (size_t)tilesX * tilesY;
    uint16_t *counts = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3);
    uint16_t *delta = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3 * K);
    uint16_t *coef = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3 * K);
    mpo_encode_tiles_fast(f, rgb, W, H, qY, qU, qV, 0, tilesX, counts, delta, coef, NULL, NULL);
    mpo_streams s;
    memset(&s, 0, sizeof s);
    s.W = W; s.H = H; s.K = K; s.bs = bs;
    const double *q[3] = {qY, qU, qV};
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) s.quant[ch][i] = (uint16_t)q[ch][i];
    for (size_t t = 0; t < tiles; t++)
        for (int ch = 0; ch < 3; ch++) {
            size_t o = t * 3 + ch;
            mpo_u16v_push(&s.lengths, counts[o]);
            for (int i = 0; i < counts[o]; i++) {
                mpo_u16v_push(&s.codes[2 * K * ch + 2 * i], delta[o * K + i]);
                mpo_u16v_push(&s.codes[2 * K * ch + 2 * i + 1], coef[o * K + i]);
            }
        }
    uint8_t *bytes = mpo_write_compressed(&s, nbytes);
    mpo_streams_free(&s);
    free(counts); free(delta); free(coef);
    return bytes;
}

/* decodeImageFast CompressedImage.cpp:837-889 */
int mpo_decode_image_fast(const uint8_t *bytes, size_t nbytes, uint8_t **rgb_out, int *Wo, int *Ho)
{
    mpo_streams s;
    if (mpo_read_compressed(bytes, nbytes, &s) != 0) { mpo_streams_free(&s); return -1; }
    const int K = s.K, bs = s.bs, W = s.W, H = s.H;
    mpo_ctx *c = mpo_ctx_create(K, bs, 0.0);
    if (!c) { mpo_streams_free(&s); return -1; }
    mpo_fast *f = mpo_fast_create(c);
    float quant[3][MPO_MAX_K];
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) quant[ch][i] = (float)s.quant[ch][i];
    uint8_t *rgb = (uint8_t *)malloc((size_t)W * H * 3);
    size_t offs[3 * MPO_MAX_K];
    memset(offs, 0, sizeof offs);
    size_t lo = 0;
    int rc = 0;
    for (int x = 0; x < W && rc == 0; x += bs)
        for (int y = 0; y < H && rc == 0; y += bs) {
            float dec[3][64];
            for (int ch = 0; ch < 3; ch++) {
                if (lo >= s.lengths.n) { rc = -1; break; }
                int cnt = s.lengths.d[lo++];
                uint16_t d[MPO_MAX_K], k[MPO_MAX_K];
                if (cnt > K) { rc = -1; break; }
                for (int i = 0; i < cnt; i++) {
                    mpo_u16v *cd = &s.codes[2 * K * ch + 2 * i], *cc = &s.codes[2 * K * ch + 2 * i + 1];
                    size_t *o = &offs[K * ch + i];
                    if (*o >= cd->n || *o >= cc->n) { rc = -1; break; }
                    d[i] = cd->d[*o];
                    k[i] = cc->d[(*o)++];
                }
                if (rc) break;
                mpo_from_coeffs_fast(f, ch, quant[ch], cnt, d, k, dec[ch]);
            }
            if (rc) break;
            for (int dx = 0; dx < bs; dx++) {
                int u = x + dx;
                for (int dy = 0; dy < bs; dy++) {
                    int v = y + dy;
                    if (u < W && v < H) {
                        uint8_t *p = rgb + 3 * ((size_t)v * W + u);
                        mpo_rgb_from_yuv((double)dec[0][dx + bs * dy], (double)dec[1][dx + bs * dy], (double)dec[2][dx + bs * dy],
                                         &p[0], &p[1], &p[2]);
                    }
                }
            }
        }
    mpo_fast_destroy(f);
    mpo_ctx_destroy(c);
    mpo_streams_free(&s);
    if (rc) { free(rgb); return rc; }
    *rgb_out = rgb; *Wo = W; *Ho = H;
    return 0;
}
