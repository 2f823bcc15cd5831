/*
 * mpo_bits.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of CompressionLib/src/BitBuffer.cpp + inc/BitBuffer.h:
 * MSB-first bit writer/reader over 64-bit words, zigzag, the reference's
 * Golomb variant, Elias-gamma and Elias-Fano sequence codes.
 */
#include "mpo.h"
#include <stdlib.h>
#include <string.h>

static void ensure_words(mpo_bits *b, size_t need)
{
    if (need <= b->cap) return;
    size_t nc = b->cap ? b->cap : 64;
    while (nc < need) nc *= 2;
    b->w = (uint64_t *)realloc(b->w, nc * sizeof(uint64_t));
    memset(b->w + b->cap, 0, (nc - b->cap) * sizeof(uint64_t));
    b->cap = nc;
}

void mpo_bits_init(mpo_bits *b)
{
    memset(b, 0, sizeof *b);
    ensure_words(b, 64);
}

void mpo_bits_free(mpo_bits *b)
{
    free(b->w);
    memset(b, 0, sizeof *b);
}

size_t mpo_bits_size(const mpo_bits *b) { return 64 * b->wword + b->wbit; }          /* BitBuffer.h:24 */
size_t mpo_bits_remaining(const mpo_bits *b)                                          /* BitBuffer.h:32 */
{
    return (b->wword - b->rword) * 64 + b->wbit - b->rbit;
}

/* BitBuffer.cpp:78-110 */
void mpo_bits_write(mpo_bits *b, uint64_t value, int width)
{
    if (width <= 0) return;          /* the reference throws for width<0 or >64; callers never do */
    ensure_words(b, b->wword + 2);
    size_t endpos = b->wbit + (size_t)width;
    uint64_t mask = (width == 64) ? ~0ULL : ((1ULL << width) - 1ULL);
    uint64_t v = value & mask;
    if (endpos <= 64) {
        b->w[b->wword] |= (endpos == 64) ? v : (v << (64 - endpos));
    } else {
        b->w[b->wword] |= v >> (endpos - 64);
        b->w[b->wword + 1] = v << (128 - endpos);
    }
    b->wbit += (size_t)width;
    if (b->wbit >= 64) {
        b->wbit -= 64;
        b->wword++;
    }
}

/* BitBuffer.cpp:118-141: the width is clipped to what remains; reads past the end give 0 */
uint64_t mpo_bits_peek(mpo_bits *b, int width)
{
    size_t rem = mpo_bits_remaining(b);
    if ((size_t)width > rem) width = (int)rem;
    if (width <= 0) return 0;
    size_t endpos = b->rbit + (size_t)width;
    uint64_t mask = (width == 64) ? ~0ULL : ((1ULL << width) - 1ULL);
    if (endpos <= 64)
        return ((endpos == 64) ? b->w[b->rword] : (b->w[b->rword] >> (64 - endpos))) & mask;
    uint64_t p1 = b->w[b->rword] << (endpos - 64);
    uint64_t p2 = b->w[b->rword + 1] >> (128 - endpos);
    return (p1 | p2) & mask;
}

/* BitBuffer.cpp:143-156 */
void mpo_bits_skip(mpo_bits *b, int width)
{
    size_t rem = mpo_bits_remaining(b);
    if ((size_t)width > rem) width = (int)rem;
    if (width <= 0) return;
    b->rbit += (size_t)width;
    if (b->rbit >= 64) {
        b->rbit -= 64;
        b->rword++;
    }
}

uint64_t mpo_bits_read(mpo_bits *b, int width)
{
    uint64_t v = mpo_bits_peek(b, width);
    mpo_bits_skip(b, width);
    return v;
}

/* BitBuffer.cpp:171-181 */
void mpo_bits_append(mpo_bits *dst, mpo_bits *src)
{
    size_t rem = mpo_bits_remaining(src);
    while (rem > 0) {
        if (rem >= 64) mpo_bits_write(dst, mpo_bits_read(src, 64), 64);
        else mpo_bits_write(dst, mpo_bits_read(src, (int)rem), (int)rem);
        rem = mpo_bits_remaining(src);
    }
}

/* BitBuffer.cpp:183-202: big-endian bytes, last partial word padded with zero bits */
uint8_t *mpo_bits_save(const mpo_bits *b, size_t *nbytes)
{
    size_t tail = (b->wbit + 7) / 8;
    size_t n = 8 * b->wword + tail;
    uint8_t *out = (uint8_t *)malloc(n ? n : 1);
    for (size_t i = 0; i < b->wword; i++)
        for (int k = 0; k < 8; k++)
            out[8 * i + k] = (uint8_t)(b->w[i] >> (56 - 8 * k));
    for (size_t k = 0; k < tail; k++)
        out[8 * b->wword + k] = (uint8_t)(b->w[b->wword] >> (8 * (7 - k)));
    *nbytes = n;
    return out;
}

/* BitBuffer.cpp:204-226 */
void mpo_bits_load(mpo_bits *b, const uint8_t *bytes, size_t start, size_t nbits)
{
    size_t words = 1 + (nbits + 63) / 64;
    ensure_words(b, words + 1);
    memset(b->w, 0, b->cap * sizeof(uint64_t));
    b->rword = b->rbit = 0;
    b->wword = nbits / 64;
    b->wbit = nbits % 64;
    for (size_t i = 0; i < b->wword; i++) {
        uint64_t v = 0;
        for (int k = 0; k < 8; k++)
            v |= (uint64_t)bytes[start + 8 * i + k] << (56 - 8 * k);
        b->w[i] = v;
    }
    uint64_t last = 0;
    for (size_t k = 0; k < (b->wbit + 7) / 8; k++)
        last |= (uint64_t)bytes[start + 8 * b->wword + k] << (8 * (7 - k));
    b->w[b->wword] = last;
}

/* BitBuffer.h:112-118 */
int32_t mpo_zigzag_dec(uint32_t x)
{
    return (int32_t)((x >> 1) ^ (uint32_t)(-((int64_t)x & 1)));
}

uint32_t mpo_zigzag_enc(int32_t x)
{
    return ((uint32_t)x << 1) ^ (uint32_t)(x >> 31);
}

static uint32_t bit_width32(uint32_t v)
{
    uint32_t n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

/* BitBuffer.cpp:228-241. b = bit_width(M), limit = 2^(b+1) - M; remainder in b
 * bits when < limit else (rem+limit) in b+1 bits -- restated as written. */
void mpo_golomb_write(uint32_t value, uint32_t M, mpo_bits *buf)
{
    uint32_t quot = (uint32_t)((int32_t)value / (int32_t)M);
    uint32_t rem = (uint32_t)((int32_t)value % (int32_t)M);
    for (uint32_t i = 0; i < quot; i++) mpo_bits_write(buf, 1, 1);
    mpo_bits_write(buf, 0, 1);
    uint32_t b = bit_width32(M);
    uint32_t limit = (1u << (b + 1)) - M;
    if (rem < limit) mpo_bits_write(buf, rem, (int)b);
    else mpo_bits_write(buf, (uint64_t)rem + limit, (int)b + 1);
}

/* BitBuffer.cpp:243-258 */
uint32_t mpo_golomb_read(uint32_t M, mpo_bits *buf)
{
    uint32_t q = 0;
    while (mpo_bits_read(buf, 1) != 0) {
        ++q;
        if (mpo_bits_remaining(buf) == 0) break;   /* the reference would spin on a truncated stream */
    }
    uint32_t b = bit_width32(M);
    uint32_t limit = (1u << (b + 1)) - M;
    uint32_t rem1 = (uint32_t)mpo_bits_read(buf, (int)b);
    uint32_t rem;
    if (rem1 < limit) rem = rem1;
    else rem = (rem1 << 1) + (uint32_t)mpo_bits_read(buf, 1) - limit;
    return q * M + rem;
}

/* BitBuffer.cpp:260-269 */
uint32_t mpo_golomb_len(uint32_t value, uint32_t M)
{
    uint32_t quot = (uint32_t)((int32_t)value / (int32_t)M);
    uint32_t rem = (uint32_t)((int32_t)value % (int32_t)M);
    uint32_t b = bit_width32(M);
    uint32_t limit = (1u << (b + 1)) - M;
    return (rem < limit) ? b + quot + 1 : b + quot + 2;
}

/* BitBuffer.cpp:271-290 */
void mpo_elias_write(uint32_t value, mpo_bits *buf)
{
    uint32_t shifted = value + 1;
    uint32_t b = bit_width32(shifted);
    mpo_bits_write(buf, 0, (int)b - 1);
    mpo_bits_write(buf, 1, 1);
    mpo_bits_write(buf, shifted, (int)b - 1);
}

uint32_t mpo_elias_read(mpo_bits *buf)
{
    uint32_t n = 0;
    while (mpo_bits_read(buf, 1) == 0) {
        ++n;
        if (mpo_bits_remaining(buf) == 0) break;
    }
    return ((1u << n) | (uint32_t)mpo_bits_read(buf, (int)n)) - 1;
}

uint32_t mpo_elias_len(uint32_t value) { return 2 * bit_width32(value + 1) - 1; }

/* BitBuffer.cpp:292-320 */
int mpo_ef_write(const uint16_t *seq, size_t n, uint16_t maxsym, mpo_bits *buf)
{
    if (n == 0) return 0;
    uint32_t m = bit_width32(maxsym);
    uint32_t nb = bit_width32((uint32_t)n);
    uint32_t lb = (m >= nb) ? m - nb : 0;
    uint16_t prev = seq[0];
    uint32_t prev_bucket = 0;
    for (size_t i = 0; i < n; i++) {
        uint16_t sym = seq[i];
        if (sym < prev) return -1;
        prev = sym;
        uint16_t bucket = (uint16_t)(sym >> lb);
        while (prev_bucket != bucket) {
            mpo_bits_write(buf, 0, 1);
            ++prev_bucket;
        }
        mpo_bits_write(buf, 1, 1);
    }
    for (size_t i = 0; i < n; i++) mpo_bits_write(buf, seq[i], (int)lb);
    return 0;
}

/* BitBuffer.cpp:322-346 */
int mpo_ef_read(uint16_t *out, size_t n, uint16_t maxsym, mpo_bits *buf)
{
    if (n == 0) return 0;
    uint32_t m = bit_width32(maxsym);
    uint32_t nb = bit_width32((uint32_t)n);
    uint32_t lb = (m >= nb) ? m - nb : 0;
    uint16_t bucket = 0;
    for (size_t i = 0; i < n; i++) {
        if (mpo_bits_remaining(buf) == 0) return -1;
        while (mpo_bits_read(buf, 1) == 0) {
            ++bucket;
            if (mpo_bits_remaining(buf) == 0) return -1;
        }
        out[i] = (uint16_t)(bucket << lb);
    }
    for (size_t i = 0; i < n; i++) out[i] |= (uint16_t)mpo_bits_read(buf, (int)lb);
    return 0;
}

/* BitBuffer.cpp:348-354 */
uint32_t mpo_ef_len(size_t n, uint16_t maxsym)
{
    if (n == 0) return 0;
    uint32_t m = bit_width32(maxsym);
    uint32_t nn = (uint32_t)n;
    uint32_t nb = bit_width32(nn);
    uint32_t lb = (m >= nb) ? m - nb : 0;
    return nn * (1 + lb) + (1u << (m - lb)) - 1;
}

void mpo_u16v_push(mpo_u16v *v, uint16_t x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 256;
        v->d = (uint16_t *)realloc(v->d, v->cap * sizeof(uint16_t));
    }
    v->d[v->n++] = x;
}

void mpo_u16v_free(mpo_u16v *v)
{
    free(v->d);
    v->d = NULL;
    v->n = v->cap = 0;
}
