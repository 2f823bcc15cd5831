/*
 * mpcodec.h -- C ABI of libmpcodec.so: the MI355X (gfx950) drop-in for the
 * CompressionLib per-tile encode path of mnesbit/ImageExperiments.
 *
 * Plain pointers and sizes only; no exceptions cross this boundary (the
 * reference throws heap-allocated std::range_error*; here every entry point
 * returns an mpc_status and mpc_last_error() carries the text).
 * Citations are relative to the reference tree (/root/reference).
 *
 * Reference interface each entry point replaces:
 *   mpc_context_create        compressed::createCompressionContext        CompressionLib/inc/CompressedImage.h:54
 *                             (createQuantizationTables CompressedImage.cpp:124, distinctLineShapes BasisSet.cpp:204,
 *                              createSegmentDictionary :299, createIntraSegmentDictionary :513)
 *   mpc_context_get/set_quant CompressionContext::{Y,U,V}.Quant           CompressedImage.h:22-36 (Compression.cpp:104-110
 *                                                                          overwrites them for "max" quality)
 *   mpc_encode_batch_device   (same, several frames per launch)
 *   mpc_encode_tiles_device   the tile loop of compressed::encodeImage    CompressedImage.cpp:535-573, i.e. per tile
 *   mpc_encode_tiles          and channel: gather + img::YUVFromRGB (misc.cpp:7) + matching::CalcMPDynamic
 *                             (MatchingPursuit.h:22, MatchingPursuit.cpp:39) over compressed::dynamicBasis
 *                             (CompressedImage.cpp:212)
 *   mpc_histogram_device      (new) per-stream symbol counts feeding huffman::huffmanEncode / golombCodeLength
 *                             (CompressedImage.cpp:359-379); the multi-GPU all-reduce operand
 *   mpc_calc_mp               matching::CalcMPDynamic on one vector       MatchingPursuit.h:22 (Compression.cpp:250 "-s" mode)
 *   mpc_write_compressed      compressed::writeCompressed (static)        CompressedImage.cpp:403
 *   mpc_read_compressed       compressed::readCompressed                  CompressedImage.cpp:635
 *   mpc_encode_image          compressed::encodeImage                     CompressedImage.h:59
 *   mpc_encode_images         (same, a sequence of frames, host and device stages overlapped)
 *   mpc_encode_image(s)_device (same, frames already in device memory)
 *   mpc_decode_image          compressed::decodeImage                     CompressedImage.h:75
 *   mpc_decode_tiles_device   matching::FromCoeffsDynamic per tile        MatchingPursuit.h:25, CompressedImage.cpp:797-831
 *   mpc_psnr                  compressed::calculatePSNR                   CompressedImage.h:57
 *   mpc_huffman_encode/decode huffman::huffmanEncode / huffmanDecode      Huffman.h:15-19
 *   mpc_rle_encode/decode     huffman::runLengthEncode / runLengthDecode  Huffman.h:12-13
 */
#ifndef MPCODEC_H
#define MPCODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_MAX_K 32
#define MPC_HIST_BINS 8192

typedef enum {
    MPC_OK = 0,
    MPC_ERR_ARGUMENT = 1,      /* bad K / block size / null pointer / geometry            */
    MPC_ERR_NO_DEVICE = 2,     /* context has no GPU (created with device < 0) or HIP failed to initialise */
    MPC_ERR_HIP = 3,           /* a HIP runtime call failed; see mpc_last_error            */
    MPC_ERR_BITSTREAM = 4,     /* "Invalid input data" / "Invalid bitstream" of the reference */
    MPC_ERR_ALLOC = 5
} mpc_status;

/* matching::BasisChoice, MatchingPursuit.h:13-17 (same layout: two u16) */
typedef struct {
    uint16_t deltaId;
    uint16_t intCoeff;
} mpc_basis_choice;

typedef struct mpc_context mpc_context;

const char* mpc_version(void);
/* text of the last failure on this thread (never NULL) */
const char* mpc_last_error(void);

/* K in 1..32, block_size must be 8 for the device path (host-only contexts accept 1..8).
 * device >= 0: HIP device ordinal, dictionary uploaded once; device < 0: host-only context
 * (dictionary/quant queries and the bitstream entry points work, encode_tiles does not). */
mpc_status mpc_context_create(int K, int block_size, double bpp_allocation, int device, mpc_context** out);
void mpc_context_destroy(mpc_context* ctx);

int mpc_context_K(const mpc_context* ctx);
int mpc_context_block_size(const mpc_context* ctx);
int mpc_context_num_base(const mpc_context* ctx);          /* 510 for 8x8 */
int mpc_context_detail_rows(const mpc_context* ctx);       /* 31622 for 8x8, per channel */
int mpc_context_device(const mpc_context* ctx);
int mpc_context_max_waves(const mpc_context* ctx);          /* resident waves of the encode kernel on the device */

/* The `...Fast` flavour of the tile path (CompressionLib/inc/MatchingPursuit.h:23,26, CompressedImage.h:67-76 -- what
 * Compression.cpp itself calls): residual, projections, quantisation, update and reconstruction in float on the dictionary
 * rounded to float.  on != 0 switches every encode / decode entry point of this context to it; the container format and the
 * entropy stage are the double path's.  PARITY UNPINNED: the reference's float results come from Eigen (absent from the
 * tree) on a dictionary built by Eigen's float eigensolver; what runs here is the reference's statements read literally in
 * float (sequential sums, products and sums rounded separately), bit-identical to oracle/mpo_fast.c and equivalent to the
 * double path in PSNR and size, not in bytes. */
mpc_status mpc_context_set_fast(mpc_context* ctx, int on);

/* How many CUs the tile encode (mpc_encode_tiles[_device], mpc_encode_batch_device) may fill: `workgroups` persistent
 * workgroups, one per CU; 0 = all of them (the default).  For callers that keep other work on the device beside it -- a
 * stripe exchange and the container jobs of the previous step (imageexperiments_amd/sharding.py: StripedEncoder.run) -- what
 * mpc_encode_images[_device] does by itself for a frame sequence: leave one CU in eight, one per shader engine, free
 * (mpc_context_max_waves / 12 = the CUs).  Not a reference interface: the reference has no device. */
mpc_status mpc_context_set_tile_encode_workgroups(mpc_context* ctx, int workgroups);
int mpc_context_is_fast(const mpc_context* ctx);

/* quant[3*K]: Y then U then V */
mpc_status mpc_context_get_quant(const mpc_context* ctx, double* quant);
mpc_status mpc_context_set_quant(mpc_context* ctx, const double* quant);

/* Copies of the host dictionary (any pointer may be NULL):
 * base[num_base*64], block_rows[num_base], detail_{y,u,v}[detail_rows*64] */
mpc_status mpc_context_get_dictionary(const mpc_context* ctx, double* base, int32_t* block_rows,
                                      double* detail_y, double* detail_u, double* detail_v);

/* ---- the hot path -------------------------------------------------------------------------------------
 * Encodes the tile rows [tile_row_begin, tile_row_end) of one RGB frame resident in device memory.
 * d_rgb: row-major, 3 bytes per pixel, `row_stride` bytes between rows (img::image<rgb>, image.h:123-131).
 * Tiles outside the image are zero filled (CompressedImage.cpp:548-552).
 * Outputs (device memory, caller allocated), tile index t = tx * rows + (ty - tile_row_begin), rows =
 * tile_row_end - tile_row_begin, i.e. the reference's x-outer / y-inner order within the stripe:
 *   d_counts [tiles][3]      u16   CalcMPDynamic's return value per channel (Y,U,V)
 *   d_choices[tiles][3][K]         records 0..count (the terminating record is written as the reference
 *                                   writes it, MatchingPursuit.cpp:50-69); entries beyond are zero
 *   d_energy [tiles][3]      f64   sum of squares of the final residual (diagnostic, not in CompressionLib)
 *   d_swept  [tiles][3]      u32   dictionary rows correlated, SURVEY 8(d) "S"
 * d_energy / d_swept may be NULL.  quant: host pointer to 3*K doubles or NULL for the context's tables.
 * stream: hipStream_t (NULL = default stream).  Asynchronous; no synchronisation, and no allocation once the
 * workspace covers the call (mpc_reserve).  waves: reserved, pass 0. */
mpc_status mpc_encode_tiles_device(mpc_context* ctx, const uint8_t* d_rgb, int width, int height, size_t row_stride,
                                   int tile_row_begin, int tile_row_end, const double* quant,
                                   uint16_t* d_counts, mpc_basis_choice* d_choices, double* d_energy, uint32_t* d_swept,
                                   int waves, void* stream);

/* Batch form: `frames` equally sized frames, `frame_stride` bytes apart; the same tile rows of every frame are
 * encoded in ONE launch (BASELINE config 4: batches of frames row-striped across GPUs).
 * Output tile index t = frame * tiles_per_stripe + tx * rows + (ty - tile_row_begin). */
mpc_status mpc_encode_batch_device(mpc_context* ctx, const uint8_t* d_rgb, int frames, size_t frame_stride,
                                   int width, int height, size_t row_stride, int tile_row_begin, int tile_row_end,
                                   const double* quant, uint16_t* d_counts, mpc_basis_choice* d_choices,
                                   double* d_energy, uint32_t* d_swept, int waves, void* stream);

/* Same with host buffers: uploads the frame, runs the kernel, copies the records back, synchronises. */
mpc_status mpc_encode_tiles(mpc_context* ctx, const uint8_t* rgb, int width, int height, size_t row_stride,
                            int tile_row_begin, int tile_row_end, const double* quant,
                            uint16_t* counts, mpc_basis_choice* choices, double* energy, uint32_t* swept);

/* d_hist[(1 + 6K)][MPC_HIST_BINS] u32 += symbol counts of the records of `tiles` tiles:
 * row 0 = lengths, row 1 + 2K*ch + 2i = deltaId at step i, +1 = intCoeff at step i. Asynchronous. */
mpc_status mpc_histogram_device(mpc_context* ctx, const uint16_t* d_counts, const mpc_basis_choice* d_choices,
                                long long tiles, uint32_t* d_hist, void* stream);

/* matching::CalcMPDynamic on one 64-vector on the device (Compression.cpp -s mode); channel 0/1/2.
 * choices[K]; *count receives the return value. Synchronous. */
mpc_status mpc_calc_mp(mpc_context* ctx, int channel, const double* quant_k, const double* input64,
                       mpc_basis_choice* choices, int* count);

/* Batch form: `count` vectors inputs[count][64] of one channel; counts[count], choices[count][K];
 * energy[count] / swept[count] optional. quant_k: K steps for that channel or NULL = context table. */
mpc_status mpc_calc_mp_batch(mpc_context* ctx, int channel, const double* quant_k, const double* inputs, int count,
                             mpc_basis_choice* choices, uint16_t* counts, double* energy, uint32_t* swept);

/* Pre-allocates the device workspace for calls of up to max_tiles tiles (3 tile-channels each).  The encode
 * entry points grow the workspace on demand, which synchronises the device; callers that must not synchronise
 * (stream capture) reserve first.  A call is cut into sub-batches that run on two internal streams (forked from and
 * joined to the caller's stream with events); at most 262144 tiles are in flight. */
mpc_status mpc_reserve(mpc_context* ctx, long long max_tiles);

/* Live timing of the dominant kernel (mp_pursuit_kernel, one launch per channel): while enabled every launch of it is
 * bracketed by HIP events on the stream it is launched on.  mpc_kernel_timing_read synchronises and returns, for
 * the launches since the last read/enable: the summed duration, their number, and (busy_ms, may be NULL) the
 * length of the union of their intervals -- launches of the two internal streams overlap.  Measurement only. */
void mpc_kernel_timing_enable(mpc_context* ctx, int on);
/* What the pursuit kernel itself counted since mpc_kernel_timing_enable(ctx, 1) or the last read (every wave adds its tallies
 * once, at exit): v_mfma_f32_16x16x32_bf16 instructions executed (16384 flop each) and tile-channel-steps.  Synchronises. */
mpc_status mpc_kernel_counters_read(mpc_context* ctx, unsigned long long* mfma_instructions, unsigned long long* tile_channel_steps);
mpc_status mpc_kernel_timing_read(mpc_context* ctx, double* total_ms, long long* launches, double* busy_ms);

/* ---- host entropy stage and container (stays on the host; bytes identical to the reference) ----------
 * Buffers returned through `uint8_t**` / `uint16_t**` are malloc'ed by the library: release with mpc_free. */
typedef struct mpc_streams mpc_streams;

void mpc_free(void* p);

/* writeCompressed (CompressedImage.cpp:403): header, DC differencing of codes[1], [2K+1], [4K+1], optional RLE,
 * Huffman-or-Golomb per stream.  quant[3*K]; lengths = 3 per tile in x-outer / y-inner tile order;
 * codes[6K] / code_lengths[6K] = the streams codes[2K*ch + 2i] (deltaId) and [+1] (intCoeff) as the encoder
 * holds them (DC coefficients NOT yet differenced). */
mpc_status mpc_write_compressed(int width, int height, int K, int block_size, const double* quant,
                                const uint16_t* lengths, size_t n_lengths, const uint16_t* const* codes,
                                const size_t* code_lengths, uint8_t** bytes, size_t* nbytes);

/* The host half of encodeImage (CompressedImage.cpp:555-575): per-tile records of a WHOLE frame, tile
 * t = tx*tiles_y + ty (what mpc_encode_tiles returns for rows [0, tiles_y)), -> container bytes. */
mpc_status mpc_assemble_streams(int width, int height, int K, int block_size, const double* quant,
                                const uint16_t* counts, const mpc_basis_choice* choices, uint8_t** bytes, size_t* nbytes);

/* Same container from records in planar order, planar[(channel * K + step) * tiles + t] (what mpc_encode_image
 * downloads after transposing on the device: each stream's records are then one contiguous run for the host). */
mpc_status mpc_assemble_planar_streams(int width, int height, int K, int block_size, const double* quant,
                                       const uint16_t* counts, const mpc_basis_choice* planar, uint8_t** bytes, size_t* nbytes);

/* Same container from streams that are already assembled (what the device's stream assembly hands the host): symbols =
 * codes[0] ++ codes[1] ++ ... ++ codes[6K-1], live symbols only, tiles in the reference's order, the three step-0 coefficient
 * streams ALREADY difference coded (CompressedImage.cpp:428-446); stream_off[6K + 1] = their boundaries in `symbols`. */
mpc_status mpc_assemble_symbol_streams(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                       const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes);

/* The same container by the route the device-side entropy stage takes (per-stream statistics -> Huffman table or Golomb M
 * and bit offsets -> codes written at their offsets), with the device's share computed on the host: a host-only check of
 * the planning half against mpc_assemble_symbol_streams. */
mpc_status mpc_assemble_symbol_streams_by_plan(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                               const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes);

/* The entropy stage of writeCompressed (CompressedImage.cpp:403-460: run-length decision, Huffman-or-Golomb choice, coding)
 * for streams the caller holds in host memory, with the per-symbol work on the device (mp_entropy.hip): run lengths,
 * histograms and first appearances are taken on the device, the host builds one code table per stream, the device writes
 * the codes.  Inputs as for mpc_assemble_symbol_streams (counts[3 * tiles] is the `lengths` stream; K and block size are the
 * context's; quant NULL = the context's tables); same bytes.  *route (optional): 0 = coded on the device, 1 = a stream was
 * outside what the device tables hold (or MPC_HOST_ENTROPY=1) and the host coded the container. */
mpc_status mpc_code_symbol_streams_device(mpc_context* ctx, int width, int height, const double* quant, const uint16_t* counts,
                                          const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes,
                                          int* route);

/* readCompressed (CompressedImage.cpp:635): parse a container; streams come back with the DC differencing
 * undone.  index -1 = lengths, 0..6K-1 = codes[index]. */
mpc_status mpc_read_compressed(const uint8_t* bytes, size_t nbytes, mpc_streams** out);
mpc_status mpc_streams_info(const mpc_streams* s, int* width, int* height, int* K, int* block_size);
mpc_status mpc_streams_quant(const mpc_streams* s, uint16_t* quant /* [3*K] */);
size_t mpc_streams_length(const mpc_streams* s, int index);
mpc_status mpc_streams_copy(const mpc_streams* s, int index, uint16_t* dst);
void mpc_streams_free(mpc_streams* s);

/* huffman::huffmanEncode / huffmanDecode (Huffman.h:15-19), runLengthEncode / runLengthDecode (:12-13) */
mpc_status mpc_huffman_encode(const uint16_t* data, size_t n, uint8_t** bytes, size_t* nbytes);
mpc_status mpc_huffman_decode(const uint8_t* bytes, size_t nbytes, uint16_t** data, size_t* n);
mpc_status mpc_rle_encode(const uint16_t* data, size_t n, uint16_t** out, size_t* n_out);
mpc_status mpc_rle_decode(const uint16_t* data, size_t n, uint16_t** out, size_t* n_out);

/* Bit-level primitives of bitbuffer:: (CompressionLib/inc/BitBuffer.h) as the entropy stage uses them; the property tests of
 * Testing/BitBufferTests.cpp:37-247 run against these (tests/test_bit_primitives.py).
 *   mpc_bits_pack / _unpack      BitBuffer::WriteBits / ReadBits + Save / Load (MSB first, widths 0..64; reads past the end give 0)
 *   mpc_zigzag_*                 zigzagEncode / zigzagDecode                    BitBuffer.h:112-118
 *   mpc_golomb_*                 golombCodeLength / writeGolombCode / readGolombCode   BitBuffer.cpp:228-269
 *   mpc_elias_fano_*             eliasFanoSequenceCodeLength / write / read           BitBuffer.cpp:292-354 */
mpc_status mpc_bits_pack(const uint64_t* values, const int* widths, size_t n, uint8_t** bytes, size_t* nbytes, size_t* nbits);
mpc_status mpc_bits_unpack(const uint8_t* bytes, size_t nbytes, const int* widths, size_t n, uint64_t* values, size_t* remaining_bits);
uint32_t mpc_zigzag_encode(int32_t x);
int32_t mpc_zigzag_decode(uint32_t x);
uint32_t mpc_golomb_length(uint32_t value, uint32_t m);
mpc_status mpc_golomb_encode(const uint32_t* values, size_t n, uint32_t m, uint8_t** bytes, size_t* nbytes, size_t* nbits);
mpc_status mpc_golomb_decode(const uint8_t* bytes, size_t nbytes, size_t n, uint32_t m, uint32_t* values, size_t* remaining_bits);
uint32_t mpc_elias_fano_length(size_t n, uint16_t max_symbol);
mpc_status mpc_elias_fano_encode(const uint16_t* sorted, size_t n, uint16_t max_symbol, uint8_t** bytes, size_t* nbytes, size_t* nbits);
mpc_status mpc_elias_fano_decode(const uint8_t* bytes, size_t nbytes, size_t n, uint16_t max_symbol, uint16_t* sorted, size_t* remaining_bits);

/* compressed::encodeImage (CompressedImage.h:59): rgb host buffer, 3*width bytes per row; quant NULL = context
 * tables.  Tile encode, stream assembly and the per-symbol work of the entropy stage (run lengths, histograms, code writing)
 * on the device; the host builds one code table per stream; only the finished container crosses PCIe (a frame whose streams do
 * not fit the device tables takes the host route for the entropy stage: same bytes). */
mpc_status mpc_encode_image(mpc_context* ctx, const uint8_t* rgb, int width, int height, const double* quant,
                            uint8_t** bytes, size_t* nbytes);

/* The same for a sequence of equally sized frames (what Compression.cpp does per input file, :117/:166), pipelined: the
 * code tables of frame n are built on the host while the device encodes frame n+1.  bytes[i] / nbytes[i] receive frame i's
 * container (each to be released with mpc_free); byte-identical to n calls of mpc_encode_image.  On failure nothing is
 * returned. */
mpc_status mpc_encode_images(mpc_context* ctx, const uint8_t* const* rgb_frames, int n_frames, int width, int height,
                             const double* quant, uint8_t** bytes, size_t* nbytes);

/* The same two with the frames already resident in device memory (3*width bytes per row, tightly packed): what bench.py
 * times. */
mpc_status mpc_encode_image_device(mpc_context* ctx, const uint8_t* d_rgb, int width, int height, const double* quant,
                                   uint8_t** bytes, size_t* nbytes);
mpc_status mpc_encode_images_device(mpc_context* ctx, const uint8_t* const* d_rgb_frames, int n_frames, int width, int height,
                                    const double* quant, uint8_t** bytes, size_t* nbytes);

/* compressed::encodeImage for a sequence of equally sized frames on SEVERAL GPUs of one node from one process (what a
 * Compression.cpp-style caller gets with MPC_DEVICES=0,1,...: dropin/compressionlib_dropin.cpp).  ctxs[0 .. n_devices): one context
 * per lane, each created on the device the lane shall use (two lanes may name the same device, not the same context), same K.
 * Every frame's tile rows are striped over the lanes (contiguous stripes, remainder to the first lanes); a step takes
 * n_devices frames, frame f of a step is owned by lane f, which pulls the other lanes' stripes of it (hipMemcpyPeerAsync), puts
 * them into the reference's tile order and produces the container.  bytes[i] / nbytes[i]: frame i's container (mpc_free),
 * byte-identical to mpc_encode_image.  n_devices == 1 is mpc_encode_images. */
mpc_status mpc_encode_images_multi(mpc_context* const* ctxs, int n_devices, const uint8_t* const* rgb_frames, int n_frames,
                                   int width, int height, const double* quant, uint8_t** bytes, size_t* nbytes);

/* Multi-GPU path, between the stripe exchange and the stream assembly: the records of tile rows [tile_row_begin, tile_row_end)
 * of a frame, in the order mpc_encode_tiles_device writes a stripe (t = tx*rows + ty_local), copied to their places in the whole
 * frame's records (t = tx*tiles_y + ty: the reference's visiting order, CompressedImage.cpp:535-537).  Asynchronous on `stream`. */
mpc_status mpc_interleave_stripe_device(mpc_context* ctx, const uint16_t* d_part_counts, const mpc_basis_choice* d_part_choices,
                                        int width, int height, int tile_row_begin, int tile_row_end, uint16_t* d_frame_counts,
                                        mpc_basis_choice* d_frame_choices, void* stream);

/* The second half of encodeImage (CompressedImage.cpp:555-575) for records that are already in device memory in whole-frame
 * order, tile t = tx*tiles_y + ty (a frame's owner in the multi-GPU path after the stripe exchange): stream assembly on the
 * device (on `stream`), live symbols to the host, entropy stage.  Synchronises `stream`. */
mpc_status mpc_records_to_container_device(mpc_context* ctx, const uint16_t* d_counts, const mpc_basis_choice* d_choices, int width,
                                           int height, const double* quant, void* stream, uint8_t** bytes, size_t* nbytes);
/* The same in three steps, for a caller that keeps the device busy meanwhile (the multi-GPU path: the next step's tile encode
 * runs while this frame's code tables are built).  slot in [0, MPC_JOB_SLOTS): one job per slot at a time; the records must
 * stay untouched until `collect` has returned (the counts are the `lengths` stream the last kernels read).
 *   begin    stream assembly and the first phase of the entropy stage enqueued on `stream`; nothing is waited for
 *   tables   waits for that, builds the code tables on the host, enqueues the second phase and the container's copy on `stream`
 *   collect  waits for the copy; the container (mpc_free)
 * A job owns its buffers: any other entry point of the context (mpc_encode_image(s), mpc_decode_image, ...) may be called between
 * `begin` and `collect`; calls on one context are serialised, not forbidden. */
#define MPC_JOB_SLOTS 6
mpc_status mpc_container_job_begin(mpc_context* ctx, int slot, const uint16_t* d_counts, const mpc_basis_choice* d_choices, int width,
                                   int height, const double* quant, void* stream);
mpc_status mpc_container_job_tables(mpc_context* ctx, int slot);
mpc_status mpc_container_job_collect(mpc_context* ctx, int slot, uint8_t** bytes, size_t* nbytes);
/* Gives a slot up whatever step its job is at (a caller that failed elsewhere between `begin` and `collect`): waits for what the
 * job has enqueued, drops its result, leaves the slot idle.  An idle slot: no-op. */
mpc_status mpc_container_job_cancel(mpc_context* ctx, int slot);

/* matching::FromCoeffsDynamic (MatchingPursuit.h:25) + img::RGBFromYUV for every tile of a frame on the device:
 * records in the reference's order (tile t = tx*tiles_y + ty, as mpc_encode_tiles returns them for the whole
 * frame), d_rgb = height*width*3 bytes.  quant: host [3*K] or NULL = context tables.  Asynchronous. */
mpc_status mpc_decode_tiles_device(mpc_context* ctx, const uint16_t* d_counts, const mpc_basis_choice* d_choices,
                                   const double* quant, int width, int height, uint8_t* d_rgb, void* stream);

/* compressed::decodeImage (CompressedImage.h:75).  The container is parsed on the host and the tiles are
 * reconstructed on ctx's device with the K and quantisation tables the stream carries (they need not equal the
 * context's; the block size must).  A context without a device gets MPC_ERR_NO_DEVICE: there is no host
 * reconstruction. */
mpc_status mpc_decode_image(const mpc_context* ctx, const uint8_t* bytes, size_t nbytes, uint8_t** rgb, int* width,
                            int* height);

/* ---- "-s" patch statistics, Compression.cpp:200-302 ----
 * The reference seeds one std::mt19937, and for every image draws `patches` origins x = rand() % (width - bs),
 * y = rand() % (height - bs), runs CalcMPDynamic on the Y, U and V patch with every quantiser 1.0 and feeds
 * intCoeff and deltaId of steps 0..count-1 to math::Stat::update (Welford, SimpleMatrix/src/covariance.cpp:5-25);
 * its text report is how Data/stats.txt and the variance tables s_varY/U/V (CompressedImage.cpp:17-122) were made.
 * Here the patches of one image are encoded in one launch of the tile encoder on ctx's device; the statistics are
 * updated on the host in the reference's order, so every double equals the reference's bit for bit. */
typedef struct mpc_patch_stats mpc_patch_stats;
mpc_status mpc_patch_stats_create(mpc_context* ctx, unsigned seed, mpc_patch_stats** out);
void mpc_patch_stats_destroy(mpc_patch_stats* s);
/* images narrower or lower than the block are skipped like Compression.cpp:233-236 (MPC_OK, nothing drawn);
 * width or height == block size is MPC_ERR_ARGUMENT (the reference divides by zero there). */
mpc_status mpc_patch_stats_add_image(mpc_patch_stats* s, const uint8_t* rgb, int width, int height, int patches);
/* out[3][2][K][5]: channel Y,U,V x {intCoeff, deltaId} x step x {N, min, max, mean, sumSq} */
mpc_status mpc_patch_stats_read(const mpc_patch_stats* s, double* out);
/* the report of Compression.cpp:275-301 as '\n'-terminated lines; release with mpc_free */
mpc_status mpc_patch_stats_report(const mpc_patch_stats* s, char** text, size_t* nbytes);
/* std::format("{}", v) of one double (shortest round-trip text), NUL terminated into buf[cap]; returns the length */
int mpc_format_double(double v, char* buf, int cap);

/* compressed::calculatePSNR (CompressedImage.h:57) */
double mpc_psnr(const uint8_t* original, const uint8_t* decoded, int width, int height);

#ifdef __cplusplus
}
#endif
#endif /* MPCODEC_H */
