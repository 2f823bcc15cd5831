/*
 * mpo.h -- ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C CPU restatement of the mnesbit/ImageExperiments CompressionLib
 * per-tile encode path (double-precision `encodeImage` / `CalcMPDynamic`
 * flavour) used only as the checker for the HIP path:
 *   - tests/ (parity), __graft_entry__.smoke(), bench.py's cpu_baseline leg.
 * Nothing under imageexperiments_amd/ may include, link or call this.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   quant tables ........ pinned by Data/r0c1de5e1t_3_5.mn header (96 values)
 *   container + entropy . pinned by the same .mn: full parse, and re-encoding
 *                         the parsed streams reproduces the file byte for byte
 *   dictionary .......... pinned (8-bit) by Data/SEG_basis.png / SEG_KLT_basis.png
 *   eigensolver, GEMV,
 *   vector ops, YUV ..... pinned bit-exactly by oracle/_ref (reference
 *                         SimpleMatrix + ImageHelper/misc compiled in place)
 *   per-tile MP outputs . no reference test or fixture holds them:
 *                         "parity unpinned" beyond the pieces above.
 *
 * All path:line citations are relative to /root/reference.
 */
#ifndef MPO_H
#define MPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPO_MAX_K 32

/* ---- symmetric eigensolver (SimpleMatrix/src/symmeigen.cpp:34-244) ---- */
/* a: n*n row-major symmetric input. vec: n*n row-major, column j = eigenvector j.
 * val: n eigenvalues (unsorted, solver order). */
void mpo_symm_eigen(const double *a, int n, double *vec, double *val);

/* createBasis (CompressionLib/src/BasisSet.cpp:118-152): rows of `basis`
 * (n*n row-major) are eigenvectors sorted by (|lambda|, index) descending and
 * sign-normalised. */
void mpo_create_basis(const double *cov, int n, double *basis);

/* ---- dictionary ---- */
typedef struct { int ax, ay, bx, by; } mpo_line;

/* covariance models BasisSet.cpp:12-24; ch = 0 Y, 1 U, 2 V */
double mpo_cov_model(int ch, double dx, double dy);

/* distinctLineShapes BasisSet.cpp:204-297; returns count, fills up to cap */
int mpo_distinct_line_shapes(int bs, mpo_line *out, int cap);

/* createSegmentDictionary BasisSet.cpp:299-380: out[nlines][bs*bs] */
void mpo_segment_dictionary(int bs, const mpo_line *lines, int nlines, double *out);

/* createIntraSegmentDictionary BasisSet.cpp:513-616: returns row count (<= bs*bs-1),
 * out[rows][bs*bs] */
int mpo_intra_segment_dictionary(int bs, const mpo_line *line, int ch, double *out);

/* createQuantizationTables CompressedImage.cpp:124-166; q[3][K] */
void mpo_quant_tables(int K, int bs, double bpp, double *qY, double *qU, double *qV);

typedef struct mpo_ctx {
    int K, bs, N;
    int nbase;              /* number of line shapes = base atoms            */
    mpo_line *lines;
    double *base;           /* [nbase][N]                                    */
    int *det_rows;          /* [nbase] rows of DetailBasis[i] (same for Y/U/V)*/
    size_t *det_off;        /* [nbase+1] row offsets into det[ch]            */
    double *det[3];         /* [det_off[nbase]][N] per channel               */
    double quant[3][MPO_MAX_K];
} mpo_ctx;

/* createCompressionContext CompressedImage.cpp:291-315 */
mpo_ctx *mpo_ctx_create(int K, int bs, double bpp);
void mpo_ctx_destroy(mpo_ctx *c);

/* ---- colour (ImageHelper/src/misc.cpp:7-36) ---- */
void mpo_yuv_from_rgb(uint8_t r, uint8_t g, uint8_t b, double *y, double *u, double *v);
void mpo_rgb_from_yuv(double y, double u, double v, uint8_t *r, uint8_t *g, uint8_t *b);

/* ---- matching pursuit (MatchingPursuit.cpp:7-74, CompressedImage.cpp:212-250) ----
 * in[N]; quant[K]; delta_id/int_coeff[K] are written for steps 0..count
 * (the terminating step's record is written as the reference does, :50-69).
 * resid[N] (optional) = residual at return; *swept (optional) = total
 * dictionary rows correlated (SURVEY 8d "S"). Returns count in [0,K]. */
int mpo_calc_mp(const mpo_ctx *c, int ch, const double *quant, const double *in,
                uint16_t *delta_id, uint16_t *int_coeff, double *resid, uint32_t *swept);

/* FromCoeffsDynamic MatchingPursuit.cpp:109-128: out[N] */
void mpo_from_coeffs(const mpo_ctx *c, int ch, const double *quant, int count,
                     const uint16_t *delta_id, const uint16_t *int_coeff, double *out);

/* gather one tile (CompressedImage.cpp:538-554): tile origin (x0,y0), rgb row-major
 * 3 B/px, out yuv[3][N] with block index dx + bs*dy, zero outside the image. */
void mpo_gather_tile(const uint8_t *rgb, int W, int H, int bs, int x0, int y0, double *yuv);

/* Per-tile records of a whole image in the reference's visiting order
 * (x outer, y inner; CompressedImage.cpp:535-537).  tile t = tx*tilesY + ty.
 * counts[t*3+ch], delta/coef[(t*3+ch)*K + i], energy[t*3+ch] = sum r^2,
 * swept[t*3+ch].  Any output pointer may be NULL.
 * tx_begin/tx_end restrict the tile columns (for bounded timing samples). */
void mpo_encode_tiles(const mpo_ctx *c, const uint8_t *rgb, int W, int H,
                      const double *qY, const double *qU, const double *qV,
                      int tx_begin, int tx_end,
                      uint16_t *counts, uint16_t *delta, uint16_t *coef,
                      double *energy, uint32_t *swept);

/* ---- the `...Fast` (float) flavour: a DEFINITION of the float mode, parity unpinned (mpo_fast.c) ---- */
typedef struct mpo_fast mpo_fast;
mpo_fast *mpo_fast_create(const mpo_ctx *c);         /* float copies of c's dictionary; c must outlive it */
void mpo_fast_destroy(mpo_fast *f);
/* CalcMPDynamicFast MatchingPursuit.cpp:76-107 */
int mpo_calc_mp_fast(const mpo_fast *f, int ch, const float *quant, const float *in,
                     uint16_t *delta_id, uint16_t *int_coeff, float *resid, uint32_t *swept);
/* FromCoeffsDynamicFast MatchingPursuit.cpp:130-147 */
void mpo_from_coeffs_fast(const mpo_fast *f, int ch, const float *quant, int count,
                          const uint16_t *delta_id, const uint16_t *int_coeff, float *out);
/* the tile loop of encodeImageFast CompressedImage.cpp:592-630; outputs as mpo_encode_tiles */
void mpo_encode_tiles_fast(const mpo_fast *f, const uint8_t *rgb, int W, int H,
                           const double *qY, const double *qU, const double *qV,
                           int tx_begin, int tx_end,
                           uint16_t *counts, uint16_t *delta, uint16_t *coef,
                           double *energy, uint32_t *swept);

/* ---- bit buffer (CompressionLib/src/BitBuffer.cpp) ---- */
typedef struct {
    uint64_t *w; size_t cap;      /* words                                  */
    size_t wword, wbit, rword, rbit;
} mpo_bits;

void mpo_bits_init(mpo_bits *b);
void mpo_bits_free(mpo_bits *b);
void mpo_bits_write(mpo_bits *b, uint64_t v, int width);
uint64_t mpo_bits_peek(mpo_bits *b, int width);
void mpo_bits_skip(mpo_bits *b, int width);
uint64_t mpo_bits_read(mpo_bits *b, int width);
size_t mpo_bits_size(const mpo_bits *b);
size_t mpo_bits_remaining(const mpo_bits *b);
void mpo_bits_append(mpo_bits *dst, mpo_bits *src);
/* Save :183 -> malloc'd bytes */
uint8_t *mpo_bits_save(const mpo_bits *b, size_t *nbytes);
void mpo_bits_load(mpo_bits *b, const uint8_t *bytes, size_t start, size_t nbits);

uint32_t mpo_zigzag_enc(int32_t x);
int32_t mpo_zigzag_dec(uint32_t x);
void mpo_golomb_write(uint32_t v, uint32_t M, mpo_bits *b);
uint32_t mpo_golomb_read(uint32_t M, mpo_bits *b);
uint32_t mpo_golomb_len(uint32_t v, uint32_t M);
void mpo_elias_write(uint32_t v, mpo_bits *b);
uint32_t mpo_elias_read(mpo_bits *b);
uint32_t mpo_elias_len(uint32_t v);
/* returns 0 ok, -1 on decreasing sequence (reference throws) */
int mpo_ef_write(const uint16_t *seq, size_t n, uint16_t maxsym, mpo_bits *b);
int mpo_ef_read(uint16_t *out, size_t n, uint16_t maxsym, mpo_bits *b);
uint32_t mpo_ef_len(size_t n, uint16_t maxsym);

/* ---- u16 vector helper ---- */
typedef struct { uint16_t *d; size_t n, cap; } mpo_u16v;
void mpo_u16v_push(mpo_u16v *v, uint16_t x);
void mpo_u16v_free(mpo_u16v *v);

/* ---- Huffman / RLE (CompressionLib/src/Huffman.cpp) ---- */
/* MPO_UMAP_MSVC: leaves enter the heap in MSVC-STL unordered_map iteration
 * order (the only toolchain the reference builds with: VS2022 .sln) --
 * this is what reproduces Data/r0c1de5e1t_3_5.mn byte for byte.
 * MPO_UMAP_SORTED: ascending symbol order (diagnostic). */
enum { MPO_UMAP_MSVC = 0, MPO_UMAP_SORTED = 1 };
void mpo_set_umap_order(int mode);

void mpo_huffman_encode(const uint16_t *data, size_t n, mpo_bits *out);
/* returns 0 ok, -1 invalid bitstream */
int mpo_huffman_decode(mpo_bits *in, mpo_u16v *out);
void mpo_rle_encode(const uint16_t *data, size_t n, mpo_u16v *out);
void mpo_rle_decode(const uint16_t *data, size_t n, mpo_u16v *out);

/* writeHuffmanOrGolomb / readHuffmanOrGolomb CompressedImage.cpp:359-401 */
void mpo_write_hog(const uint16_t *data, size_t n, mpo_bits *out);
int mpo_read_hog(mpo_bits *in, size_t length, mpo_u16v *out);

/* ---- container (CompressedImage.cpp:403-460, 635-707) ---- */
typedef struct {
    int W, H, K, bs;
    uint16_t quant[3][MPO_MAX_K];
    mpo_u16v lengths;               /* 3 per tile, x-outer/y-inner          */
    mpo_u16v codes[6 * MPO_MAX_K];  /* codes[2K*ch + 2i (+1)]               */
} mpo_streams;

void mpo_streams_free(mpo_streams *s);
/* DC-diffs codes[1],[2K+1],[4K+1] in place like the reference, returns bytes */
uint8_t *mpo_write_compressed(mpo_streams *s, size_t *nbytes);
/* returns 0 ok; codes come back with the DC diff undone (as :690-705) */
int mpo_read_compressed(const uint8_t *bytes, size_t nbytes, mpo_streams *s);

/* encodeImage CompressedImage.cpp:521-576 */
uint8_t *mpo_encode_image(const mpo_ctx *c, const uint8_t *rgb, int W, int H,
                          const double *qY, const double *qU, const double *qV,
                          size_t *nbytes);
/* decodeImage :783-835; rgb_out = W*H*3 malloc'd; returns 0 ok */
int mpo_decode_image(const uint8_t *bytes, size_t nbytes, uint8_t **rgb_out, int *W, int *H);

/* encodeImageFast :578-633 / decodeImageFast :837-889 */
uint8_t *mpo_encode_image_fast(const mpo_fast *f, const uint8_t *rgb, int W, int H,
                               const double *qY, const double *qU, const double *qV, size_t *nbytes);
int mpo_decode_image_fast(const uint8_t *bytes, size_t nbytes, uint8_t **rgb_out, int *W, int *H);

/* calculatePSNR :343-357 */
double mpo_psnr(const uint8_t *a, const uint8_t *b, int W, int H);

double mpo_variance_constant(int ch, int i);
double mpo_decay_constant(int ch);

/* "-s" patch statistics (Compression.cpp:200-302; Welford update SimpleMatrix/src/covariance.cpp:5-25).
 * Parity unpinned beyond the pursuit itself: the images Data/stats.txt was made from are not in the reference tree. */
typedef struct mpo_patch_stats mpo_patch_stats;
mpo_patch_stats *mpo_patch_stats_create(int K, uint32_t seed);
void mpo_patch_stats_destroy(mpo_patch_stats *p);
void mpo_patch_stats_add_image(mpo_patch_stats *p, const mpo_ctx *c, const uint8_t *rgb, int W, int H, int patches);
void mpo_patch_stats_read(const mpo_patch_stats *p, double *out /* [3][2][K][5]: N,min,max,mean,sumSq */);

/* synthetic frame generator of BASELINE.md 3 (std::mt19937 restated) */
void mpo_synth_frame(uint8_t *rgb, int W, int H, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif
