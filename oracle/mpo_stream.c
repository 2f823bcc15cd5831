/*
 * mpo_stream.c -- ORACLE (test infrastructure, not product code).
 *
 * Restatement of the host entropy stage and container:
 *   CompressionLib/src/Huffman.cpp:46-163 (encode), :173-244 (decode),
 *                                   :246-310 (u16 run-length code)
 *   CompressionLib/src/CompressedImage.cpp:359-401 (Huffman-or-Golomb),
 *                                   :403-460 (writeCompressed),
 *                                   :635-707 (readCompressed),
 *                                   :521-576 (encodeImage), :783-835 (decodeImage)
 *
 * Huffman tie-breaking.  huffmanEncode seeds its priority_queue by iterating a
 * std::unordered_map<uint32_t,...> (Huffman.cpp:47,72-81), so equal-frequency
 * ties are broken by the container's iteration order, which the C++ standard
 * leaves to the implementation.  The reference builds only with Visual Studio
 * 2022 (README.md:9, ImageExperiments.sln), and the committed fixture
 * Data/r0c1de5e1t_3_5.mn was written by that build.  The MSVC STL is not part
 * of /root/reference; its published algorithm (<xhash>, VS2019 16.3+ layout)
 * is restated here:
 *   - std::hash<unsigned> = 64-bit FNV-1a over the key's 4 bytes;
 *   - one doubly linked list of all elements + per-bucket [lo,hi] iterators,
 *     bucket = hash & (buckets-1), 8 buckets initially, max_load_factor 1;
 *   - a new key whose bucket is empty is appended at the END of the list,
 *     otherwise it is linked in FRONT of its bucket's first element;
 *   - growth when size+1 > buckets: x8 while buckets < 512, afterwards the next
 *     power of two >= size+1; rehash walks the list in order and moves each
 *     element in front of its new bucket's first element (or leaves it in
 *     place if the bucket is still empty).
 * std::priority_queue = std::push_heap/pop_heap (sift-up; hole-to-bottom then
 * sift-up), identical in MSVC STL and libstdc++.
 * tests/test_oracle_golden.py proves the restatement: decoding the .mn fixture
 * and re-encoding its streams reproduces all 3 698 188 bytes.
 */
#include "mpo.h"
#include <stdlib.h>
#include <string.h>

static int g_umap_mode = MPO_UMAP_MSVC;
void mpo_set_umap_order(int mode) { g_umap_mode = mode; }

/* ------------------------------------------------------------------ */
/* MSVC-STL unordered_map<uint32_t, T> iteration-order emulation        */
/* ------------------------------------------------------------------ */
typedef struct {
    uint32_t *key;
    int *next, *prev;       /* node 0 = list head / end()               */
    int nodes, cap;
    int *lo, *hi;           /* per bucket, 0 = empty                    */
    size_t buckets, size;
} msvc_umap;

static uint64_t fnv1a_u32(uint32_t k)
{
    uint64_t h = 14695981039346656037ULL;
    for (int i = 0; i < 4; i++) {
        h ^= (uint64_t)((k >> (8 * i)) & 0xFFu);
        h *= 1099511628211ULL;
    }
    return h;
}

static void um_init(msvc_umap *m)
{
    memset(m, 0, sizeof *m);
    m->cap = 64;
    m->key = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)m->cap);
    m->next = (int *)malloc(sizeof(int) * (size_t)m->cap);
    m->prev = (int *)malloc(sizeof(int) * (size_t)m->cap);
    m->next[0] = m->prev[0] = 0;
    m->nodes = 1;
    m->buckets = 8;
    m->lo = (int *)calloc(m->buckets, sizeof(int));
    m->hi = (int *)calloc(m->buckets, sizeof(int));
}

static void um_free(msvc_umap *m)
{
    free(m->key); free(m->next); free(m->prev); free(m->lo); free(m->hi);
}

static void um_link_before(msvc_umap *m, int node, int before)
{
    int after = m->prev[before];
    m->next[node] = before;
    m->prev[node] = after;
    m->next[after] = node;
    m->prev[before] = node;
}

static void um_unlink(msvc_umap *m, int node)
{
    m->next[m->prev[node]] = m->next[node];
    m->prev[m->next[node]] = m->prev[node];
}

/* _Forced_rehash */
static void um_rehash(msvc_umap *m, size_t want)
{
    size_t nb = 1;
    while (nb < want) nb <<= 1;
    free(m->lo); free(m->hi);
    m->buckets = nb;
    m->lo = (int *)calloc(nb, sizeof(int));
    m->hi = (int *)calloc(nb, sizeof(int));
    int ins = m->next[0];
    while (ins != 0) {
        int nxt = m->next[ins];
        size_t b = (size_t)(fnv1a_u32(m->key[ins]) & (nb - 1));
        if (m->lo[b] == 0) {
            m->lo[b] = m->hi[b] = ins;
        } else {
            um_unlink(m, ins);
            um_link_before(m, ins, m->lo[b]);
            m->lo[b] = ins;
        }
        ins = nxt;
    }
}

/* _Find_last: where a new (absent) key would be linked */
static int um_insert_point(const msvc_umap *m, uint32_t key)
{
    size_t b = (size_t)(fnv1a_u32(key) & (m->buckets - 1));
    if (m->hi[b] == 0) return 0;        /* empty bucket -> before end() */
    return m->lo[b];                    /* otherwise in front of the bucket */
}

/* operator[] on an absent key (_Try_emplace); returns the node id */
static int um_insert_new(msvc_umap *m, uint32_t key)
{
    if (m->nodes == m->cap) {
        m->cap *= 2;
        m->key = (uint32_t *)realloc(m->key, sizeof(uint32_t) * (size_t)m->cap);
        m->next = (int *)realloc(m->next, sizeof(int) * (size_t)m->cap);
        m->prev = (int *)realloc(m->prev, sizeof(int) * (size_t)m->cap);
    }
    int node = m->nodes++;
    m->key[node] = key;
    /* _Check_rehash_required_1: max_load_factor (1.0f) < (size+1)/buckets */
    if (1.0f < (float)(m->size + 1) / (float)m->buckets) {
        size_t req = m->size + 1;                        /* _Min_load_factor_buckets */
        if (req < 8) req = 8;
        size_t grow = (m->buckets < 512 && m->buckets * 8 >= req) ? m->buckets * 8 : req;
        um_rehash(m, grow);
    }
    int before = um_insert_point(m, key);
    size_t b = (size_t)(fnv1a_u32(key) & (m->buckets - 1));
    int after = m->prev[before];
    um_link_before(m, node, before);
    if (m->lo[b] == 0) { m->lo[b] = m->hi[b] = node; }
    else if (m->lo[b] == before) { m->lo[b] = node; }
    else if (m->hi[b] == after) { m->hi[b] = node; }
    m->size++;
    return node;
}

/* ------------------------------------------------------------------ */
/* Huffman encode  (Huffman.cpp:46-163)                                 */
/* ------------------------------------------------------------------ */
#define PSEUDO_EOF 0xFFFFFFFFu

typedef struct {
    uint8_t depth;
    size_t freq;
    int head, tail;         /* chain of leaf ids (children, in order)    */
} hnode;

/* NodeCompare :21-25  a.frequency > b.frequency */
static int node_after(const hnode *a, const hnode *b) { return a->freq > b->freq; }

/* std::push_heap on h[0..n) where h[n-1] is the new element */
static void heap_sift_up(hnode *h, int hole, int top, hnode val)
{
    for (int idx = (hole - 1) >> 1; top < hole && node_after(&h[idx], &val); idx = (hole - 1) >> 1) {
        h[hole] = h[idx];
        hole = idx;
    }
    h[hole] = val;
}

/* std::pop_heap + pop_back on h[0..*n) */
static void heap_pop(hnode *h, int *n)
{
    int len = *n;
    if (len >= 2) {
        hnode val = h[len - 1];
        int bottom = len - 1;
        int hole = 0, idx = 0;
        const int last_nonleaf = (bottom - 1) >> 1;
        while (idx < last_nonleaf) {
            idx = 2 * idx + 2;
            if (node_after(&h[idx], &h[idx - 1])) --idx;
            h[hole] = h[idx];
            hole = idx;
        }
        if (idx == last_nonleaf && bottom % 2 == 0) {
            h[hole] = h[bottom - 1];
            hole = bottom - 1;
        }
        heap_sift_up(h, hole, 0, val);
    }
    *n = len - 1;
}

typedef struct { uint32_t symbol; uint8_t len; uint32_t code; } hentry;

static int canonical_cmp(const void *pa, const void *pb)   /* CanonicalSorter :27-35 */
{
    const hentry *a = (const hentry *)pa, *b = (const hentry *)pb;
    if (a->len != b->len) return a->len < b->len ? -1 : 1;
    if (a->symbol != b->symbol) return a->symbol < b->symbol ? -1 : 1;
    return 0;
}

static uint8_t bit_width_u32(uint32_t v) { uint8_t n = 0; while (v) { n++; v >>= 1; } return n; }

void mpo_huffman_encode(const uint16_t *data, size_t n, mpo_bits *buffer)
{
    /* leaf id = insertion order; node_of[sym] = leaf id + 1 */
    int *node_of = (int *)calloc(65536, sizeof(int));
    size_t leaf_cap = 256, nleaf = 0;
    uint32_t *leaf_sym = (uint32_t *)malloc(sizeof(uint32_t) * leaf_cap);
    size_t *leaf_freq = (size_t *)malloc(sizeof(size_t) * leaf_cap);
    msvc_umap um;
    um_init(&um);
    uint8_t symbol_bits = 1;
    for (size_t i = 0; i < n; i++) {                       /* :49-65 */
        uint32_t s = data[i];
        uint8_t bw = bit_width_u32(s);
        if (bw > symbol_bits) symbol_bits = bw;
        if (!node_of[s]) {
            if (nleaf + 2 > leaf_cap) {
                leaf_cap *= 2;
                leaf_sym = (uint32_t *)realloc(leaf_sym, sizeof(uint32_t) * leaf_cap);
                leaf_freq = (size_t *)realloc(leaf_freq, sizeof(size_t) * leaf_cap);
            }
            leaf_sym[nleaf] = s;
            leaf_freq[nleaf] = 0;
            node_of[s] = (int)++nleaf;
            um_insert_new(&um, s);                         /* umap node id == leaf id + 1 */
        }
        ++leaf_freq[node_of[s] - 1];
    }
    leaf_sym[nleaf] = PSEUDO_EOF;                          /* :66-72 */
    leaf_freq[nleaf] = 0;
    const int eof_leaf = (int)nleaf++;
    um_insert_new(&um, PSEUDO_EOF);

    uint8_t *len = (uint8_t *)calloc(nleaf, 1);
    len[eof_leaf] = (uint8_t)(n == 0 ? 1 : 0);
    int *chain = (int *)malloc(sizeof(int) * nleaf);       /* next leaf in a node's child list */
    hnode *heap = (hnode *)malloc(sizeof(hnode) * (nleaf + 1));
    int hn = 0;

    /* :73-82 leaves enter the queue in map iteration order */
    if (g_umap_mode == MPO_UMAP_MSVC) {
        for (int nd = um.next[0]; nd != 0; nd = um.next[nd]) {
            int leaf = nd - 1;
            chain[leaf] = -1;
            hnode v = {0, leaf_freq[leaf], leaf, leaf};
            hn++;
            heap_sift_up(heap, hn - 1, 0, v);
        }
    } else {
        hentry *tmp = (hentry *)malloc(sizeof(hentry) * nleaf);
        for (size_t i = 0; i < nleaf; i++) { tmp[i].symbol = leaf_sym[i]; tmp[i].len = 0; tmp[i].code = (uint32_t)i; }
        qsort(tmp, nleaf, sizeof(hentry), canonical_cmp);
        for (size_t i = 0; i < nleaf; i++) {
            int leaf = (int)tmp[i].code;
            chain[leaf] = -1;
            hnode v = {0, leaf_freq[leaf], leaf, leaf};
            hn++;
            heap_sift_up(heap, hn - 1, 0, v);
        }
        free(tmp);
    }
    /* :83-104 */
    while (hn > 1) {
        hnode a = heap[0];
        heap_pop(heap, &hn);
        hnode b = heap[0];
        heap_pop(heap, &hn);
        uint8_t depth = a.depth > b.depth ? a.depth : b.depth;
        chain[a.tail] = b.head;
        for (int c = a.head; c != -1; c = chain[c]) ++len[c];
        ++depth;
        hnode m = {depth, a.freq + b.freq, a.head, b.tail};
        hn++;
        heap_sift_up(heap, hn - 1, 0, m);
    }
    /* :105-109 */
    hentry *ent = (hentry *)malloc(sizeof(hentry) * nleaf);
    for (size_t i = 0; i < nleaf; i++) { ent[i].symbol = leaf_sym[i]; ent[i].len = len[i]; ent[i].code = 0; }
    qsort(ent, nleaf, sizeof(hentry), canonical_cmp);
    uint8_t max_len = heap[0].depth > 1 ? heap[0].depth : 1;
    mpo_bits_write(buffer, max_len, 8);
    /* :111-134 canonical codes + per-length counts */
    uint16_t *group = (uint16_t *)malloc(sizeof(uint16_t) * 300);
    int ngroup = 0;
    uint8_t prev_len = 0;
    uint32_t code = 0, count = 0;
    uint32_t *code_of = (uint32_t *)calloc(65536, sizeof(uint32_t));
    uint8_t *len_of = (uint8_t *)calloc(65536, 1);
    uint32_t eof_code = 0; uint8_t eof_len = 0;
    for (size_t i = 0; i < nleaf; i++) {
        hentry *e = &ent[i];
        if (e->len != prev_len) {
            if (prev_len != 0) {
                mpo_bits_write(buffer, (uint16_t)count, 16);
                group[ngroup++] = (uint16_t)count;
            }
            for (uint8_t l = (uint8_t)(prev_len + 1); l < e->len; ++l) {
                mpo_bits_write(buffer, 0, 16);
                group[ngroup++] = 0;
            }
            count = 0;
            code = code << (e->len - prev_len);
        }
        e->code = code;
        ++code;
        ++count;
        prev_len = e->len;
        if (e->symbol == PSEUDO_EOF) { eof_code = e->code; eof_len = e->len; }
        else { code_of[e->symbol] = e->code; len_of[e->symbol] = e->len; }
    }
    mpo_bits_write(buffer, (uint16_t)count, 16);
    group[ngroup++] = (uint16_t)count;
    mpo_bits_write(buffer, symbol_bits, 8);
    /* :137-156 symbol table, Elias-Fano or raw per length group */
    uint16_t mask = (uint16_t)((1u << symbol_bits) - 1u);
    size_t pos = 0;
    int gi = 0;
    uint16_t *range = (uint16_t *)malloc(sizeof(uint16_t) * (nleaf + 1));
    while (pos < nleaf) {
        uint16_t rl = group[gi++];
        for (uint16_t k = 0; k < rl; k++) range[k] = (uint16_t)((uint16_t)ent[pos + k].symbol & mask);
        if (mpo_ef_len(rl, mask) < (uint32_t)rl * (uint32_t)symbol_bits)
            mpo_ef_write(range, rl, mask, buffer);
        else
            for (uint16_t k = 0; k < rl; k++) mpo_bits_write(buffer, range[k], symbol_bits);
        pos += rl;
    }
    /* :157-162 payload + pseudo-EOF */
    for (size_t i = 0; i < n; i++) mpo_bits_write(buffer, code_of[data[i]], len_of[data[i]]);
    mpo_bits_write(buffer, eof_code, eof_len);

    free(range); free(len_of); free(code_of); free(group); free(ent); free(heap);
    free(chain); free(len); free(leaf_freq); free(leaf_sym); free(node_of);
    um_free(&um);
}

/* Huffman.cpp:173-244 */
int mpo_huffman_decode(mpo_bits *in, mpo_u16v *decoded)
{
    uint8_t max_len = (uint8_t)mpo_bits_read(in, 8);
    uint16_t counts[256];
    uint16_t total = 0;
    for (uint8_t l = 1; l <= max_len && l != 0; ++l) {
        uint16_t c = (uint16_t)mpo_bits_read(in, 16);
        total = (uint16_t)(total + c);
        counts[l - 1] = c;
    }
    uint8_t symbol_bits = (uint8_t)mpo_bits_read(in, 8);
    uint16_t mask = (uint16_t)((1u << symbol_bits) - 1u);
    uint16_t *symbols = (uint16_t *)malloc(sizeof(uint16_t) * ((size_t)total + 1));
    size_t ns = 0;
    for (int l = 0; l < max_len; l++) {
        uint16_t rl = counts[l];
        if (mpo_ef_len(rl, mask) < (uint32_t)rl * (uint32_t)symbol_bits) {
            if (mpo_ef_read(symbols + ns, rl, mask, in) != 0) { free(symbols); return -1; }
        } else {
            for (int i = 0; i < rl; i++) symbols[ns + i] = (uint16_t)mpo_bits_read(in, symbol_bits);
        }
        ns += rl;
    }
    /* canonical table :205-230: first code / first symbol index per length */
    uint32_t first_code[257], first_idx[257];
    uint32_t code = 0, idx = 0;
    uint8_t prev = 0;
    for (int l = 1; l <= max_len; l++) {
        if (counts[l - 1]) {
            code = code << (l - prev);
            first_code[l] = code;
            first_idx[l] = idx;
            code += counts[l - 1];
            idx += counts[l - 1];
            prev = (uint8_t)l;
        } else {
            first_code[l] = 0; first_idx[l] = 0;
        }
    }
    if (total == 0 || symbols[total - 1] != mask) { free(symbols); return -1; }   /* :218-221 */
    /* :231-243 bit-serial decode until the pseudo-EOF (last table entry) */
    uint32_t acc = 0;
    int bits = 0;
    while (mpo_bits_remaining(in) > 0) {
        acc = (acc << 1) | (uint32_t)mpo_bits_read(in, 1);
        ++bits;
        if (bits <= max_len && counts[bits - 1] && acc >= first_code[bits] &&
            acc - first_code[bits] < counts[bits - 1]) {
            uint32_t si = first_idx[bits] + (acc - first_code[bits]);
            if (si == (uint32_t)total - 1) { free(symbols); return 0; }
            mpo_u16v_push(decoded, symbols[si]);
            acc = 0;
            bits = 0;
        }
    }
    free(symbols);
    return -1;
}

/* Huffman.cpp:246-279 */
void mpo_rle_encode(const uint16_t *data, size_t n, mpo_u16v *out)
{
    uint16_t prev = 0, count = 0;
    int new_run = 1;
    for (size_t i = 0; i < n; i++) {
        uint16_t val = data[i];
        if (val == prev && !new_run) {
            ++count;
            if (count == 1) {
                mpo_u16v_push(out, val);
            } else if (count >= 0x8000) {
                mpo_u16v_push(out, (uint16_t)(count - 1));
                count = 0;
                new_run = 1;
            }
        } else {
            new_run = 0;
            if (count > 0) {
                mpo_u16v_push(out, (uint16_t)(count - 1));
                count = 0;
            }
            prev = val;
            mpo_u16v_push(out, val);
        }
    }
    if (count > 0) mpo_u16v_push(out, (uint16_t)(count - 1));
}

/* Huffman.cpp:281-310 */
void mpo_rle_decode(const uint16_t *data, size_t n, mpo_u16v *out)
{
    uint16_t prev = 0;
    int pending = 0, new_run = 1;
    for (size_t i = 0; i < n; i++) {
        uint16_t val = data[i];
        if (pending) {
            for (uint16_t k = 0; k < val; ++k) mpo_u16v_push(out, prev);
            pending = 0;
            new_run = 1;
        } else {
            mpo_u16v_push(out, val);
            if (val == prev && !new_run) pending = 1;
            new_run = 0;
            prev = val;
        }
    }
}

/* CompressedImage.cpp:359-390 */
void mpo_write_hog(const uint16_t *data, size_t n, mpo_bits *out)
{
    mpo_bits tmp;
    mpo_bits_init(&tmp);
    mpo_huffman_encode(data, n, &tmp);
    size_t best = mpo_bits_remaining(&tmp);
    int best_m = -1;
    int m = 1;
    while (m < 2048) {
        size_t est = 16;
        for (size_t i = 0; i < n; i++) est += mpo_golomb_len(data[i], (uint32_t)m);
        if (est < best) { best = est; best_m = m; }
        if ((m & 1) == 1) ++m;
        else m = (m << 1) - 1;
    }
    if (best_m == -1) {
        mpo_bits_write(out, 0, 1);
        mpo_bits_append(out, &tmp);
    } else {
        mpo_bits_write(out, 1, 1);
        mpo_bits_write(out, (uint16_t)best_m, 16);
        for (size_t i = 0; i < n; i++) mpo_golomb_write(data[i], (uint32_t)best_m, out);
    }
    mpo_bits_free(&tmp);
}

/* CompressedImage.cpp:392-401 */
int mpo_read_hog(mpo_bits *in, size_t length, mpo_u16v *out)
{
    if (mpo_bits_read(in, 1) == 0) return mpo_huffman_decode(in, out);
    uint32_t m = (uint32_t)mpo_bits_read(in, 16);
    if (m == 0) return -1;
    for (size_t i = 0; i < length; i++) mpo_u16v_push(out, (uint16_t)mpo_golomb_read(m, in));
    return 0;
}

void mpo_streams_free(mpo_streams *s)
{
    mpo_u16v_free(&s->lengths);
    for (int i = 0; i < 6 * MPO_MAX_K; i++) mpo_u16v_free(&s->codes[i]);
}

static void dc_diff(mpo_u16v *v)                       /* :428-446 */
{
    int32_t prev = 0;
    for (size_t i = 0; i < v->n; i++) {
        int32_t diff = (int32_t)v->d[i] - prev;
        prev = (int32_t)v->d[i];
        v->d[i] = (uint16_t)mpo_zigzag_enc(diff);
    }
}

static void dc_undiff(mpo_u16v *v)                     /* :690-705 */
{
    int32_t acc = 0;
    for (size_t i = 0; i < v->n; i++) {
        acc = mpo_zigzag_dec((uint32_t)v->d[i]) + acc;
        v->d[i] = (uint16_t)acc;
    }
}

#define MAGIC 0x4D4E3234u

/* CompressedImage.cpp:403-460 */
uint8_t *mpo_write_compressed(mpo_streams *s, size_t *nbytes)
{
    const int K = s->K;
    mpo_bits out;
    mpo_bits_init(&out);
    mpo_bits_write(&out, MAGIC, 32);
    mpo_bits_write(&out, (uint32_t)s->W, 32);
    mpo_bits_write(&out, (uint32_t)s->H, 32);
    mpo_bits_write(&out, (uint8_t)K, 8);
    mpo_bits_write(&out, (uint8_t)s->bs, 8);
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) mpo_bits_write(&out, s->quant[ch][i], 16);
    dc_diff(&s->codes[1]);
    dc_diff(&s->codes[2 * K + 1]);
    dc_diff(&s->codes[4 * K + 1]);
    mpo_write_hog(s->lengths.d, s->lengths.n, &out);
    for (int i = 0; i < 6 * K; i++) {
        mpo_u16v comp = {0, 0, 0};
        mpo_rle_encode(s->codes[i].d, s->codes[i].n, &comp);
        if (comp.n + 4 < s->codes[i].n) {
            mpo_bits_write(&out, 1, 1);
            mpo_bits_write(&out, (uint32_t)comp.n, 32);
            mpo_write_hog(comp.d, comp.n, &out);
        } else {
            mpo_bits_write(&out, 0, 1);
            mpo_write_hog(s->codes[i].d, s->codes[i].n, &out);
        }
        mpo_u16v_free(&comp);
    }
    uint8_t *bytes = mpo_bits_save(&out, nbytes);
    mpo_bits_free(&out);
    return bytes;
}

/* CompressedImage.cpp:635-707 */
int mpo_read_compressed(const uint8_t *bytes, size_t nbytes, mpo_streams *s)
{
    memset(s, 0, sizeof *s);
    mpo_bits in;
    mpo_bits_init(&in);
    mpo_bits_load(&in, bytes, 0, 8 * nbytes);
    int rc = -1;
    if ((uint32_t)mpo_bits_read(&in, 32) != MAGIC) goto done;
    s->W = (int)(uint32_t)mpo_bits_read(&in, 32);
    s->H = (int)(uint32_t)mpo_bits_read(&in, 32);
    s->K = (int)mpo_bits_read(&in, 8);
    s->bs = (int)mpo_bits_read(&in, 8);
    if (s->K < 1 || s->K > MPO_MAX_K || s->bs < 1 || s->bs > 8) goto done;
    const int K = s->K;
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) s->quant[ch][i] = (uint16_t)mpo_bits_read(&in, 16);
    size_t tiles = (size_t)((s->W + s->bs - 1) / s->bs) * (size_t)((s->H + s->bs - 1) / s->bs);
    size_t patch = 3 * tiles;
    if (mpo_read_hog(&in, patch, &s->lengths) != 0) goto done;
    for (int i = 0; i < 6 * K; i++) {
        if (mpo_bits_read(&in, 1) == 1) {
            size_t clen = (size_t)(uint32_t)mpo_bits_read(&in, 32);
            mpo_u16v comp = {0, 0, 0};
            if (mpo_read_hog(&in, clen, &comp) != 0) { mpo_u16v_free(&comp); goto done; }
            mpo_rle_decode(comp.d, comp.n, &s->codes[i]);
            mpo_u16v_free(&comp);
        } else {
            size_t layer = (size_t)(i / 2) / (size_t)K, depth = (size_t)(i / 2) % (size_t)K;
            size_t layer_len = 0;
            for (size_t o = 0; o < s->lengths.n / 3; o++)
                if (s->lengths.d[3 * o + layer] > (uint16_t)depth) ++layer_len;
            if (mpo_read_hog(&in, layer_len, &s->codes[i]) != 0) goto done;
        }
    }
    dc_undiff(&s->codes[1]);
    dc_undiff(&s->codes[2 * K + 1]);
    dc_undiff(&s->codes[4 * K + 1]);
    rc = 0;
done:
    mpo_bits_free(&in);
    return rc;
}

/* CompressedImage.cpp:521-576 */
uint8_t *mpo_encode_image(const mpo_ctx *c, const uint8_t *rgb, int W, int H,
                          const double *qY, const double *qU, const double *qV,
                          size_t *nbytes)
{
    const int bs = c->bs, K = c->K;
    const int tilesX = (W + bs - 1) / bs, tilesY = (H + bs - 1) / bs;
    const size_t tiles = (size_t)tilesX * tilesY;
    uint16_t *counts = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3);
    uint16_t *delta = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3 * K);
    uint16_t *coef = (uint16_t *)malloc(sizeof(uint16_t) * tiles * 3 * K);
    mpo_encode_tiles(c, rgb, W, H, qY, qU, qV, 0, tilesX, counts, delta, coef, NULL, NULL);
    mpo_streams s;
    memset(&s, 0, sizeof s);
    s.W = W; s.H = H; s.K = K; s.bs = bs;
    const double *q[3] = {qY, qU, qV};
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) s.quant[ch][i] = (uint16_t)q[ch][i];
    for (size_t t = 0; t < tiles; t++)                 /* t already runs x-outer / y-inner */
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

/* CompressedImage.cpp:783-835 */
int mpo_decode_image(const uint8_t *bytes, size_t nbytes, uint8_t **rgb_out, int *Wo, int *Ho)
{
    mpo_streams s;
    if (mpo_read_compressed(bytes, nbytes, &s) != 0) { mpo_streams_free(&s); return -1; }
    const int K = s.K, bs = s.bs, W = s.W, H = s.H, N = bs * bs;
    mpo_ctx *c = mpo_ctx_create(K, bs, 0.0);
    if (!c) { mpo_streams_free(&s); return -1; }
    double quant[3][MPO_MAX_K];
    for (int ch = 0; ch < 3; ch++)
        for (int i = 0; i < K; i++) quant[ch][i] = (double)s.quant[ch][i];
    uint8_t *rgb = (uint8_t *)malloc((size_t)W * H * 3);
    size_t offs[3 * MPO_MAX_K];
    memset(offs, 0, sizeof offs);
    size_t lo = 0;
    int rc = 0;
    for (int x = 0; x < W && rc == 0; x += bs)
        for (int y = 0; y < H && rc == 0; y += bs) {
            double dec[3][64];
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
                mpo_from_coeffs(c, ch, quant[ch], cnt, d, k, dec[ch]);
            }
            if (rc) break;
            for (int dx = 0; dx < bs; dx++) {
                int u = x + dx;
                for (int dy = 0; dy < bs; dy++) {
                    int v = y + dy;
                    if (u < W && v < H) {
                        uint8_t *p = rgb + 3 * ((size_t)v * W + u);
                        mpo_rgb_from_yuv(dec[0][dx + bs * dy], dec[1][dx + bs * dy], dec[2][dx + bs * dy],
                                         &p[0], &p[1], &p[2]);
                    }
                }
            }
        }
    mpo_ctx_destroy(c);
    mpo_streams_free(&s);
    if (rc) { free(rgb); return rc; }
    *rgb_out = rgb; *Wo = W; *Ho = H;
    (void)N;
    return 0;
}
