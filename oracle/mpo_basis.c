/*
 * mpo_basis.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of the dictionary construction of
 *   CompressionLib/src/BasisSet.cpp  (double-precision variants)
 * and of the context factory CompressedImage.cpp:291-315.
 */
#include "mpo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* BasisSet.cpp:12-24 (abs on double = fabs under the reference's toolchain) */
double mpo_cov_model(int ch, double dx, double dy)
{
    switch (ch) {
    case 0:
        return 3817.7299999999996 * exp(-1.48854e-05 * dx * dx + -1.7273e-05 * dy * dy)
             + 657.8100000000001 * exp(-0.0436057 * fabs(dx) + -0.050844400000000005 * fabs(dy));
    case 1:
        return 241.49 * exp(-0.00134755 * fabs(dx) + -0.00147572 * fabs(dy));
    default:
        return 371.87199999999996 * exp(-0.00147084 * fabs(dx) + -0.0015265799999999998 * fabs(dy));
    }
}

/* BasisSet.cpp:188-190.  Integer numerator, sqrt of an integer; a degenerate
 * line gives 0/0 = NaN and the callers' ">= 0.0" is then false. */
static double side_dist(int l1x, int l1y, int l2x, int l2y, int px, int py)
{
    int num = (l2x - l1x) * (l1y - py) - (l1x - px) * (l2y - l1y);
    int den = (l2x - l1x) * (l2x - l1x) + (l2y - l1y) * (l2y - l1y);
    return (double)num / sqrt((double)den);
}

/* A shape = bs*bs booleans, index x + y*bs (:215).  Packed so that unsigned
 * comparison of keys equals ShapeComparator (:192-202): the first differing
 * position decides and `true` sorts first. Requires bs*bs <= 64. */
static uint64_t shape_key(const unsigned char *bits, int n)
{
    uint64_t k = 0;
    for (int i = 0; i < n; i++)
        if (!bits[i])
            k |= 1ULL << (63 - i);
    return k;
}

typedef struct { uint64_t key; mpo_line line; } shape_entry;

typedef struct { shape_entry *e; int n, cap; } shape_map;

static int map_find(const shape_map *m, uint64_t key)
{
    for (int i = 0; i < m->n; i++)
        if (m->e[i].key == key)
            return i;
    return -1;
}

static void map_put(shape_map *m, uint64_t key, mpo_line l)
{
    if (m->n == m->cap) {
        m->cap = m->cap ? 2 * m->cap : 256;
        m->e = (shape_entry *)realloc(m->e, sizeof(shape_entry) * (size_t)m->cap);
    }
    m->e[m->n].key = key;
    m->e[m->n].line = l;
    m->n++;
}

static int entry_cmp(const void *a, const void *b)
{
    uint64_t ka = ((const shape_entry *)a)->key, kb = ((const shape_entry *)b)->key;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

static void fill_bits(int bs, mpo_line l, unsigned char *bits, unsigned char *inv)
{
    for (int x = 0; x < bs; x++)
        for (int y = 0; y < bs; y++) {
            unsigned char b = side_dist(l.ax, l.ay, l.bx, l.by, x, y) >= 0.0;
            bits[x + y * bs] = b;
            if (inv)
                inv[x + y * bs] = !b;
        }
}

/* try-insert of :249-257 / :283-291: skip when the mask or its inverse is present */
static void put_unless_either(shape_map *m, int bs, mpo_line l)
{
    unsigned char bits[64], inv[64];
    fill_bits(bs, l, bits, inv);
    uint64_t k = shape_key(bits, bs * bs), ki = shape_key(inv, bs * bs);
    if (map_find(m, k) < 0 && map_find(m, ki) < 0)
        map_put(m, k, l);
}

/* BasisSet.cpp:204-297 */
int mpo_distinct_line_shapes(int bs, mpo_line *out, int cap)
{
    shape_map m = {0, 0, 0};
    if (bs * bs > 64)
        return -1;
    /* :207-225 strict horizontals / verticals; only the mask itself is tested */
    for (int side = -1; side < bs + 1; side++) {
        mpo_line horiz = {0, side, bs, side};
        mpo_line vert = {side, 0, side, bs};
        unsigned char hb[64], vb[64];
        fill_bits(bs, horiz, hb, NULL);
        fill_bits(bs, vert, vb, NULL);
        uint64_t kh = shape_key(hb, bs * bs), kv = shape_key(vb, bs * bs);
        if (map_find(&m, kh) < 0)
            map_put(&m, kh, horiz);
        if (map_find(&m, kv) < 0)
            map_put(&m, kv, vert);
    }
    /* :226-259 lines from the far top edge */
    for (int s1 = -bs; s1 < 2 * bs; s1++)
        for (int s2 = -bs; s2 < 2 * bs; s2++) {
            mpo_line a = {s1, -bs, -bs, s2};
            mpo_line b = {s1, -bs, s2, bs};
            mpo_line c = {s1, -bs, bs, s2};
            put_unless_either(&m, bs, a);
            put_unless_either(&m, bs, b);
            put_unless_either(&m, bs, c);
        }
    /* :260-293 lines from the far right edge (s1b is unused in the reference;
     * its third candidate repeats the second) */
    for (int s1 = -bs; s1 < 2 * bs; s1++)
        for (int s2 = -bs; s2 < 2 * bs; s2++) {
            mpo_line a = {bs, s1, -bs, s2};
            mpo_line b = {bs, s1, s2, bs};
            put_unless_either(&m, bs, a);
            put_unless_either(&m, bs, b);
            put_unless_either(&m, bs, b);
        }
    qsort(m.e, (size_t)m.n, sizeof(shape_entry), entry_cmp);   /* std::map order :294-296 */
    int n = m.n;
    for (int i = 0; i < n && i < cap; i++)
        out[i] = m.e[i].line;
    free(m.e);
    return n;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* BasisSet.cpp:299-380 */
void mpo_segment_dictionary(int bs, const mpo_line *lines, int nlines, double *out)
{
    const int N = bs * bs;
    const double sigma = 1.0;
    const int kw = (int)(1 + sigma * 6);
    const int kh = kw / 2;
    double gauss[49];
    for (int dx = -kh; dx <= kh; dx++)
        for (int dy = -kh; dy <= kh; dy++)
            gauss[(dx + kh) + kw * (dy + kh)] = exp(-(double)(dx * dx + dy * dy) / (2.0 * sigma * sigma));
    double *zoom = (double *)malloc(sizeof(double) * 4 * (size_t)N);
    for (int i = 0; i < nlines; i++) {
        double *row = out + (size_t)i * N;
        int all_set = 1, all_clear = 1;
        int ax = lines[i].ax * 2, ay = lines[i].ay * 2, bx = lines[i].bx * 2, by = lines[i].by * 2;
        for (int x = 0; x < 2 * bs; x++)
            for (int y = 0; y < 2 * bs; y++) {
                int bit = side_dist(ax, ay, bx, by, x, y) >= 0.0;
                if (bit) all_clear = 0; else all_set = 0;
                zoom[x + 2 * bs * y] = bit ? +1.0 : -1.0;
            }
        double total = 0.0;
        for (int x = 0; x < bs; x++)
            for (int y = 0; y < bs; y++) {
                double weight = 0.0, tot = 0.0;
                for (int dx = -kh; dx <= kh; dx++) {
                    int u = clampi(2 * x + dx, 0, 2 * bs - 1);
                    for (int dy = -kh; dy <= kh; dy++) {
                        int v = clampi(2 * y + dy, 0, 2 * bs - 1);
                        double c = gauss[(dx + kh) + kw * (dy + kh)];
                        weight += c;
                        tot += c * zoom[u + 2 * bs * v];
                    }
                }
                double value = tot / weight;
                total += value;
                row[x + bs * y] = value;
            }
        double mean = total / (double)N;
        double sumsq = 0.0;
        for (int j = 0; j < N; j++) {
            double value = row[j];
            if (!(all_set || all_clear)) {
                value -= mean;
                row[j] = value;
            }
            sumsq += value * value;
        }
        double norm = sqrt(sumsq);
        for (int j = 0; j < N; j++)
            if (sumsq != 0.0)
                row[j] = row[j] / norm;
    }
    free(zoom);
}

/* one side of :559-613: KLT vector `vecrow` spread over the side's pixels,
 * mean-removed over the side, L2-normalised. guard = the reference only tests
 * sumSq != 0 for side 2 (:607). */
static void emit_side_row(int bs, const int *px, const int *py, int cnt,
                          const double *vecrow, int guard, double *dst)
{
    const int N = bs * bs;
    memset(dst, 0, sizeof(double) * (size_t)N);
    double mean = 0.0;
    for (int j = 0; j < cnt; j++) {
        double v = vecrow[j];
        mean += v;
        dst[px[j] + bs * py[j]] = v;
    }
    mean /= (double)cnt;
    double ss = 0.0;
    for (int j = 0; j < cnt; j++) {
        double v = dst[px[j] + bs * py[j]];
        v -= mean;
        ss += v * v;
        dst[px[j] + bs * py[j]] = v;
    }
    ss = sqrt(ss);
    for (int j = 0; j < N; j++) {
        double v = dst[j];
        if (!guard || ss != 0.0)
            v /= ss;
        dst[j] = v;
    }
}

/* BasisSet.cpp:513-616 */
int mpo_intra_segment_dictionary(int bs, const mpo_line *line, int ch, double *out)
{
    const int N = bs * bs;
    int x1[64], y1[64], x2[64], y2[64], n1 = 0, n2 = 0;
    for (int x = 0; x < bs; x++)
        for (int y = 0; y < bs; y++) {
            if (side_dist(line->ax, line->ay, line->bx, line->by, x, y) >= 0.0) {
                x1[n1] = x; y1[n1] = y; n1++;
            } else {
                x2[n2] = x; y2[n2] = y; n2++;
            }
        }
    double *cov1 = (double *)malloc(sizeof(double) * (size_t)(n1 * n1 + 1));
    double *cov2 = (double *)malloc(sizeof(double) * (size_t)(n2 * n2 + 1));
    double *b1 = (double *)malloc(sizeof(double) * (size_t)(n1 * n1 + 1));
    double *b2 = (double *)malloc(sizeof(double) * (size_t)(n2 * n2 + 1));
    for (int i = 0; i < n1; i++)
        for (int j = 0; j < n1; j++)
            cov1[i * n1 + j] = mpo_cov_model(ch, (double)(x1[i] - x1[j]), (double)(y1[i] - y1[j]));
    for (int i = 0; i < n2; i++)
        for (int j = 0; j < n2; j++)
            cov2[i * n2 + j] = mpo_cov_model(ch, (double)(x2[i] - x2[j]), (double)(y2[i] - y2[j]));
    mpo_create_basis(cov1, n1, b1);
    mpo_create_basis(cov2, n2, b2);
    int c1 = n1 - 1 > 0 ? n1 - 1 : 0;
    int c2 = n2 - 1 > 0 ? n2 - 1 : 0;
    int rows = 0, most = c1 > c2 ? c1 : c2;
    for (int i = 0; i < most; i++) {
        if (i < c1) {
            emit_side_row(bs, x1, y1, n1, b1 + (size_t)(i + 1) * n1, 0, out + (size_t)rows * N);
            rows++;
        }
        if (i < c2) {
            emit_side_row(bs, x2, y2, n2, b2 + (size_t)(i + 1) * n2, 1, out + (size_t)rows * N);
            rows++;
        }
    }
    free(cov1); free(cov2); free(b1); free(b2);
    return rows;
}

/* CompressedImage.cpp:291-315 */
mpo_ctx *mpo_ctx_create(int K, int bs, double bpp)
{
    if (K < 1 || K > MPO_MAX_K || bs < 1 || bs * bs > 64)
        return NULL;
    mpo_ctx *c = (mpo_ctx *)calloc(1, sizeof(mpo_ctx));
    c->K = K; c->bs = bs; c->N = bs * bs;
    mpo_quant_tables(K, bs, bpp, c->quant[0], c->quant[1], c->quant[2]);
    int cap = 4096;
    c->lines = (mpo_line *)malloc(sizeof(mpo_line) * (size_t)cap);
    c->nbase = mpo_distinct_line_shapes(bs, c->lines, cap);
    c->base = (double *)malloc(sizeof(double) * (size_t)c->nbase * c->N);
    mpo_segment_dictionary(bs, c->lines, c->nbase, c->base);
    c->det_rows = (int *)malloc(sizeof(int) * (size_t)c->nbase);
    c->det_off = (size_t *)malloc(sizeof(size_t) * (size_t)(c->nbase + 1));
    double *tmp = (double *)malloc(sizeof(double) * (size_t)c->N * c->N);
    for (int ch = 0; ch < 3; ch++) {
        size_t used = 0, room = (size_t)c->nbase * (size_t)c->N;
        c->det[ch] = (double *)malloc(sizeof(double) * room * c->N);
        for (int i = 0; i < c->nbase; i++) {
            int r = mpo_intra_segment_dictionary(bs, &c->lines[i], ch, tmp);
            c->det_rows[i] = r;
            c->det_off[i] = used;
            memcpy(c->det[ch] + used * c->N, tmp, sizeof(double) * (size_t)r * c->N);
            used += (size_t)r;
        }
        c->det_off[c->nbase] = used;
    }
    free(tmp);
    return c;
}

void mpo_ctx_destroy(mpo_ctx *c)
{
    if (!c) return;
    free(c->lines); free(c->base); free(c->det_rows); free(c->det_off);
    for (int ch = 0; ch < 3; ch++) free(c->det[ch]);
    free(c);
}
