/*
 * mpo_eigen.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of the reference's symmetric eigensolver
 *   SimpleMatrix/src/symmeigen.cpp:34-244   (Householder tridiagonalisation,
 *   accumulation of the transformations, QL with implicit shifts)
 * and of basis::createBasis
 *   CompressionLib/src/BasisSet.cpp:118-152.
 * Every floating-point operation is kept in the reference's evaluation order
 * (compile with -ffp-contract=off); oracle/_ref checks this file bit-for-bit
 * against the reference's own object code.
 */
#include "mpo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* symmeigen.cpp:9-10 -- the reference's own abs template: (x > 0 ? x : -x) */
static double mag(double x) { return x > 0 ? x : -x; }

/* symmeigen.cpp:18-32 */
static double hyp(double a, double b)
{
    if (mag(a) > mag(b)) {
        double r = b / a;
        return mag(a) * sqrt(1.0 + r * r);
    }
    if (b != 0.0) {
        double r = a / b;
        return mag(b) * sqrt(1.0 + r * r);
    }
    return 0.0;
}

/* symmeigen.cpp:47-124: reduce to tridiagonal form, last row first.
 * V holds the matrix on entry; d = diagonal scratch, e = off-diagonal. */
static void tridiagonalise(double *V, int n, double *d, double *e)
{
    for (int c = 0; c < n; c++)
        d[c] = V[(n - 1) * n + c];

    for (int i = n - 1; i > 0; i--) {
        double scale = 0.0, h = 0.0;
        for (int j = 0; j < i; j++)
            scale = scale + mag(d[j]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; j++) {
                d[j] = V[(i - 1) * n + j];
                V[i * n + j] = 0.0;
                V[j * n + i] = 0.0;
            }
        } else {
            for (int j = 0; j < i; j++) {
                double t = d[j] / scale;
                d[j] = t;
                h += t * t;
            }
            double f = d[i - 1];
            double g = sqrt(h);
            if (f > 0)
                g = -g;
            e[i] = scale * g;
            h = h - f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; j++)
                e[j] = 0.0;
            for (int j = 0; j < i; j++) {
                f = d[j];
                V[j * n + i] = f;
                g = e[j] + V[j * n + j] * f;
                for (int k = j + 1; k <= i - 1; k++) {
                    g += V[k * n + j] * d[k];
                    e[k] += V[k * n + j] * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; j++) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            double hh = f / (h + h);
            for (int j = 0; j < i; j++)
                e[j] -= hh * d[j];
            for (int j = 0; j < i; j++) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; k++)
                    V[k * n + j] -= (f * e[k] + g * d[k]);
                d[j] = V[(i - 1) * n + j];
                V[i * n + j] = 0.0;
            }
        }
        d[i] = h;
    }
}

/* symmeigen.cpp:126-160: build the orthogonal matrix of the reduction */
static void accumulate(double *V, int n, double *d, double *e)
{
    for (int i = 0; i < n - 1; i++) {
        V[(n - 1) * n + i] = V[i * n + i];
        V[i * n + i] = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int j = 0; j <= i; j++)
                d[j] = V[j * n + i + 1] / h;
            for (int j = 0; j <= i; j++) {
                double g = 0.0;
                for (int k = 0; k <= i; k++)
                    g += V[k * n + i + 1] * V[k * n + j];
                for (int k = 0; k <= i; k++)
                    V[k * n + j] -= g * d[k];
            }
        }
        for (int j = 0; j <= i; j++)
            V[j * n + i + 1] = 0.0;
    }
    for (int i = 0; i < n; i++) {
        d[i] = V[(n - 1) * n + i];
        V[(n - 1) * n + i] = 0.0;
    }
    V[(n - 1) * n + n - 1] = 1.0;
    e[0] = 0.0;
}

/* symmeigen.cpp:162-243: QL iterations with implicit shift */
static void ql_implicit(double *V, int n, double *d, double *e)
{
    double shift = 0.0, piv = 0.0;
    const double eps = 1.0E-20;
    for (int i = 1; i < n; i++)
        e[i - 1] = e[i];
    e[n - 1] = 0.0;
    for (int i = 0; i < n; i++) {
        double cand = mag(d[i]) + mag(e[i]);
        piv = (piv < cand) ? cand : piv;          /* std::max(piv, cand) */
        int l = i;
        while (l < n) {
            if (mag(e[l]) <= eps * piv)
                break;
            l++;
        }
        if (l > i) {
            do {
                double f = d[i];
                double g = (d[i + 1] - f) / (2.0 * e[i]);
                double r = hyp(g, 1.0);
                if (g < 0)
                    r = -r;
                d[i] = e[i] / (g + r);
                d[i + 1] = e[i] * (g + r);
                double dnext = d[i + 1];
                double h = f - d[i];
                for (int j = i + 2; j < n; j++)
                    d[j] -= h;
                shift = shift + h;
                g = d[l];
                double enext = e[i + 1];
                double c1 = 1.0, c2 = 1.0, c3 = 1.0, s1 = 0.0, s2 = 0.0;
                for (int j = l - 1; j >= i; j--) {
                    c3 = c2;
                    c2 = c1;
                    s2 = s1;
                    f = c1 * e[j];
                    h = c1 * g;
                    r = hyp(g, e[j]);
                    e[j + 1] = s1 * r;
                    s1 = e[j] / r;
                    c1 = g / r;
                    g = c1 * d[j] - s1 * f;
                    d[j + 1] = h + s1 * (c1 * f + s1 * d[j]);
                    for (int k = 0; k < n; k++) {
                        h = V[k * n + j + 1];
                        V[k * n + j + 1] = s1 * V[k * n + j] + c1 * h;
                        V[k * n + j] = c1 * V[k * n + j] - s1 * h;
                    }
                }
                g = -s1 * s2 * c3 * enext * e[i] / dnext;
                e[i] = s1 * g;
                d[i] = c1 * g;
            } while (mag(e[i]) > eps * piv);
        }
        d[i] = d[i] + shift;
        e[i] = 0.0;
    }
}

void mpo_symm_eigen(const double *a, int n, double *vec, double *val)
{
    if (n <= 0)
        return;
    double *e = (double *)calloc((size_t)n, sizeof(double));
    memcpy(vec, a, sizeof(double) * (size_t)n * (size_t)n);
    memset(val, 0, sizeof(double) * (size_t)n);
    tridiagonalise(vec, n, val, e);
    accumulate(vec, n, val, e);
    ql_implicit(vec, n, val, e);
    free(e);
}

/* ---- createBasis: BasisSet.cpp:118-152 ---- */
typedef struct { double key; int idx; } keyed;

/* std::greater<std::pair<double,int>> :134 */
static int keyed_desc(const void *pa, const void *pb)
{
    const keyed *a = (const keyed *)pa, *b = (const keyed *)pb;
    if (a->key > b->key) return -1;
    if (a->key < b->key) return 1;
    if (a->idx > b->idx) return -1;
    if (a->idx < b->idx) return 1;
    return 0;
}

void mpo_create_basis(const double *cov, int n, double *basis)
{
    if (n <= 0)
        return;
    double *vec = (double *)malloc(sizeof(double) * (size_t)n * (size_t)n);
    double *val = (double *)malloc(sizeof(double) * (size_t)n);
    keyed *order = (keyed *)malloc(sizeof(keyed) * (size_t)n);
    mpo_symm_eigen(cov, n, vec, val);
    for (int i = 0; i < n; i++) {
        order[i].key = fabs(val[i]);           /* :132 abs(eigenValues[i]) */
        order[i].idx = i;
    }
    qsort(order, (size_t)n, sizeof(keyed), keyed_desc);
    for (int i = 0; i < n; i++) {
        double *row = basis + (size_t)i * n;
        int col = order[i].idx;
        for (int k = 0; k < n; k++)            /* GetColumn, mathmatrix.cpp:158 */
            row[k] = vec[(size_t)k * n + col];
        for (int j = 0; j < n; j++) {          /* :139-148 sign convention */
            if (fabs(row[j]) > 1E-10) {
                if (row[j] < 0.0)
                    for (int k = 0; k < n; k++)
                        row[k] = -row[k];
                break;
            }
        }
    }
    free(order);
    free(val);
    free(vec);
}
