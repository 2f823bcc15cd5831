// mp_device.h -- product: device-side data layout and launchers of the MI355X (gfx950) quantized matching-pursuit tile
// encoder and of the stages behind it.
//
//   mp_pursuit.hip   the PERSISTENT pursuit kernel (the product's path): one launch runs all K steps of every tile-channel of a
//                    batch on chip -- LDS-resident dictionary, split-bf16 MFMA screen with a proven bound, Gram-updated detail
//                    blocks, exact evaluation of the survivors in the reference's arithmetic (PursuitArgs, launch_pursuit)
//   mp_streams.hip   stream assembly: the records -> the container's 6K symbol streams, live symbols only (StreamArgs)
//   mp_entropy.hip   the per-symbol work of the entropy stage: run lengths, histograms, first appearances, code writing (EntropyArgs)
//   mp_kernels.hip   the decoder, the symbol histogram, and the product's own cross-check of the tile encoder: a STEP-SYNCHRONOUS
//                    pursuit that correlates every row exactly (a short sequence of kernels per MP step over the active
//                    tile-channels: init, fill, base sweep, detail sweep, finish, update), behind MPC_PATH=steps / MPC_FILTER=0
// The step-synchronous kernels' layout follows (Workspace, enqueue_pursuit); the persistent kernel's is further down.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mpc {

constexpr int kHistBins = 8192;
constexpr int kMaxDeviceK = 32;
constexpr int kMaxParts = 8;            // base sweep can be split over up to 8 atom ranges per tile-channel group
constexpr int kChunkItems = 64;         // items (tile-channels) a wave processes per loaded detail block
constexpr int kNumBuckets = 3 * 512;    // (channel, block) buckets, block < 510
constexpr int kBaseFilterTiles = 32;    // 16-row tiles of the filter copy of the base dictionary (512 rows)
constexpr int kBlockFilterTiles = 4;    // ... of one detail block (62 | 63 rows, padded to 64)
constexpr int kMaxRowParts = 4;         // a detail block's 62/63 rows can be split over up to 4 waves

struct DictDevice {
    const double* base;              // [base_rows_padded + 1][64] row-major, zero rows after num_base
    int num_base;                    // 510
    int base_rows_padded;
    const double* detail;            // [3][detail_rows][64] row-major (+1 zero row at the very end)
    const float* base32;             // the same two, rounded to float (the `...Fast` flavour)
    const float* detail32;
    long long detail_rows;           // rows per channel (31 622)
    const int32_t* block_rows;       // [num_base]
    int block0_rows;                 // block_rows[0] (host copy)
    const int32_t* block_row_off;    // [num_base+1]
};

// Per-batch device workspace (all device pointers). cap = max tile-channels per batch.
struct Workspace {
    int cap;
    double* r;                       // [cap][64] residuals
    double* part_val;                // [cap][kMaxParts] best projection of each base atom range
    int* part_idx;                   // [cap][kMaxParts]
    double* cand0_val;               // [cap] best projection on DetailBasis[0] (if unlocked)
    int* cand0_row;                  // [cap]
    float* approx_max;               // [cap] filter pass: largest approximate projection over base rows + block 0
    int* prev_id;                    // [cap]
    int* nblk;                       // [cap] entries in blk_list
    int* extra_rows;                 // [cap] rows appended after the base part (duplicates included)
    unsigned* swept;                 // [cap]
    uint16_t* blk_list;              // [cap][32] chosen base atoms in order: bits 0-8 atom, 9-14 rows of its block, 15 = repeat
    int* item_slot;                  // [cap][32] bucket slot of each blk_list entry for the current step
    int* out_index;                  // [cap] record index in the low 30 bits, channel in the top 2
    int* act[2][3];                  // [cap] active tile-channels per channel (ping-pong)
    unsigned* counters;              // [16]: [cur*3+ch] = active count; 6 = chunk cursor; 7 = n_chunks; 8 = n_items
    unsigned* bucket_count[2];       // [kNumBuckets] items per (channel, block) for this / the next step
    unsigned* bucket_cursor;         // [kNumBuckets]
    int* chunks;                     // [max_chunks][4] = bucket, begin, end, 0
    int* items;                      // [max_items] tile-channel of each item
    double* cand_val;                // [max_items][kMaxRowParts] best projection of the item's block, per row range
    int* cand_row;                   // [max_items][kMaxRowParts]
    double* upd_coeff;               // [cap] finish -> update: coefficient of the chosen atom (0 = no update)
    int* upd_sel;                    // [cap] ~idx for a base atom, row index into `detail` otherwise
    long long max_items;
    int max_chunks;
};

struct FrameInput {
    const uint8_t* rgb;              // device, row-major, 3 B/pixel (img::image<rgb>, image.h:123-131)
    int width, height;
    long long row_stride;
    int frames;
    long long frame_stride;
    int tile_row_begin, tile_rows, tiles_x;
    int out_tile_rows;               // 0: records in stripe order (tile tx * tile_rows + ty - tile_row_begin); > 0: in the order of a
                                     // whole frame of this many tile rows (tx * out_tile_rows + ty): stripes encoded one by one
                                     // land where one launch over the frame would put them
    // vector mode (matching::CalcMPDynamic on caller vectors): vec_in != nullptr
    const double* vec_in;            // [n][64]
    int vec_channel;
};

struct Outputs {
    uint16_t* counts;                // [records]
    uint32_t* choices;               // [records][K]  lo16 = deltaId, hi16 = intCoeff
    double* energy;                  // [records] or null
    uint32_t* swept;                 // [records] or null
};

struct HistParams {
    const uint16_t* counts;          // [tiles][3]
    const uint32_t* choices;         // [tiles][3][K]
    long long tiles;
    int K;
    uint32_t* hist;                  // [(1 + 6K)][8192]
};

// Enqueue the whole K-step pursuit for the tile-channels [tc_begin, tc_begin + n) of the input on `stream`.
// Tile-channel numbering: tc = unit*3 + ch, unit = frame*tiles_per_frame_stripe + tx*tile_rows + (ty - begin)
// (vector mode: tc = vector index, channel fixed).  Returns hipError_t as int.  No host synchronisation,
// no allocation: graph-capturable.
// side_stream/fork_event/join_event: optional second stream (+ two events) on which each step's residual update runs
// beside the next step's bucket + fill kernels; nullptr = everything on `stream`.
// base_events: optional array of 2*K hipEvent_t (as void*) recorded right before / after each base-sweep launch
// (live per-kernel timing for bench.py's roofline); nullptr = none.
int enqueue_pursuit(const DictDevice& dict, const Workspace& ws, const FrameInput& in, const Outputs& out,
                    const double* quant_dev, int K, long long tc_begin, int n, int parts, int row_parts, int sweep_waves,
                    void* stream, void** base_events, void* side_stream, void* fork_event, void* join_event);

int launch_histogram(const HistParams& p, void* stream);

// Decoder: FromCoeffsDynamic + RGBFromYUV for whole tiles.
struct DecodeParams {
    const uint16_t* counts;          // [tiles][3], tile t = tx*tiles_y + ty (the reference's visiting order)
    const uint32_t* choices;         // [tiles][3][K]
    const double* quant;             // [3][K] (device)
    int K;
    int width, height, tiles_x, tiles_y;
    uint8_t* rgb;                    // [height][width][3] (device)
    int* error_flag;                 // set to 1 if a record indexes outside its dynamic dictionary
    int fast;                        // != 0: FromCoeffsDynamicFast (float)
};
int launch_decode(const DictDevice& dict, const DecodeParams& p, void* stream);

// ---- persistent pursuit (mp_pursuit.hip): one launch runs all K steps of every tile-channel of a batch ----
constexpr int kMaxPairs = 32;           // (tile-channel, unlocked block other than DetailBasis[0]) pairs a tile-channel can hold (< K)

struct PursuitArgs {
    // dictionary
    const double* base;              // [512][64]
    const double* detail[3];         // per channel [detail_rows][64]
    const uint16_t* base_tiles;      // [kBaseFilterTiles][2048], k order 1 (host_dictionary.h: filter_tiles)
    const uint16_t* block_tiles[3];  // per channel [num_base][kBlockFilterTiles][2048], k order 1
    const float* gram[3];            // per channel [num_base + detail_rows][gram_stride]
    long long gram_stride;           // num_base * 64
    const int32_t* block_rows;
    const int32_t* block_row_off;
    const double* quant;             // [3][K] (device)
    int K, num_base, rows0;
    // float flavour (`...Fast`): the same dictionary rounded to float, row layout as base / detail; fast != 0 selects it
    const float* base32;
    const float* detail32[3];
    int fast;
    // One launch covers all channels with `workgroups` workgroups (one per CU).  Every WAVE works on one channel at a time: luma
    // first, and once a channel's queue is dry the wave moves on to the next channel that has work (mp_pursuit.hip: channel
    // switch), so a channel's last long tile-channels drain beside the next channel's work instead of beside idle SIMDs.
    int workgroups;
    // input: tile mode (rgb) or vector mode (vec_in != nullptr: CalcMPDynamic on caller vectors of channel vec_channel)
    const uint8_t* rgb;
    int width, height;
    long long row_stride, frame_stride;
    int tile_row_begin, tile_rows, tiles_x;
    int out_tile_rows;               // FrameInput::out_tile_rows
    int rgb_aligned8;                // rgb, row_stride and frame_stride are multiples of 8: whole tiles are read as 8-byte words
    const double* vec_in;
    int vec_channel;
    long long n_tc[3];               // tile-channels of each channel (tiles of the stripe x frames; vector mode: only vec_channel's is not 0)
    unsigned* queue;                 // [3] next tile-channel of each channel (zero before the launch)
    // per-wave scratch for the pairs (sizes: pursuit_scratch_*): a wave's own, whatever channel it works on
    float* pair_p;
    unsigned* pair_meta;
    float* pair_e;
    Outputs out;
    unsigned long long* stats;       // [2] += executed MFMA instructions, tile-channel-steps (one atomic per wave at exit); may be null
    unsigned long long* debug;       // diagnostic builds (-DMPC_STAMPS) only: 24 phase-cycle / event counters; else null
};

int launch_pursuit(const PursuitArgs& args, void* stream);       // grid = args.workgroups
size_t pursuit_scratch_floats(int workgroups);
size_t pursuit_scratch_meta(int workgroups);
size_t pursuit_scratch_bounds(int workgroups);
int pursuit_units_per_workgroup();
// Gram table of one channel (see mp_pursuit.hip); shadow[detail_rows]: 1 = row left out of the filter copy
int launch_gram(const double* base, const double* detail, const int32_t* block_rows, const int32_t* block_row_off,
                const uint8_t* shadow, float* gram, int num_base, int n_sel, long long stride, void* stream);

// ---- device-side stream assembly (mp_streams.hip) ----
struct StreamArgs {
    const uint16_t* counts;          // [tiles][3]
    const uint32_t* choices;         // [tiles][3][K]
    long long tiles;
    int K;
    unsigned* block_live;            // [stream_workspace_words(tiles, K)] scratch
    unsigned* sizes;                 // [3][K] out: symbols per (channel, step)
    unsigned long long* stream_off;  // [6K + 1] out: stream boundaries in `symbols` (container order)
    uint16_t* symbols;               // [2 * 3 * K * tiles] worst case; out: the 6K streams back to back
    uint16_t* dc_tmp;                // [3][tiles] scratch
};
size_t stream_workspace_words(long long tiles, int K);
int launch_stream_assembly(const StreamArgs& a, void* stream);
// the decoder's way back: counts + the 6K streams (in `symbols`, container order, DC coefficients already summed) -> records
// [tiles][3][K], dead steps zero.  Uses a.counts, a.symbols (read), a.block_live and a.sizes (scratch); not a.stream_off / a.dc_tmp
int launch_stream_gather(const StreamArgs& a, uint32_t* choices, void* stream);
// records of the tile rows [row_begin, row_begin + rows) in stripe order -> their places in the whole frame's records
int launch_interleave_stripe(const uint16_t* part_counts, const uint32_t* part_choices, int tiles_x, int tiles_y, int row_begin, int rows,
                             int K, uint16_t* frame_counts, uint32_t* frame_choices, void* stream);

// ---- device-side entropy stage (mp_entropy.hip): everything that touches every symbol of the 1 + 6K streams ----
constexpr int kEntBlock = 4096;         // symbols per scan block
constexpr int kEntHistSpan = 4;         // scan blocks per histogram span and range of symbol values (ent_hist_kernel)
constexpr int kEntMaxStreams = 6 * kMaxDeviceK + 1;

struct EntStream {                      // one per stream; the device fills the first part, the host the second
    unsigned long long raw_off;         // first symbol of the stream in `symbols` (stream 0 = `lengths`: the counts array)
    unsigned n;                         // symbols of the stream as assembled
    unsigned blk_begin;                 // first scan block
    unsigned rle_size;                  // symbols runLengthEncode emits (Huffman.cpp:246-279)
    unsigned shorter;                   // 1 = the run-length coded stream is what gets coded (CompressedImage.cpp:450)
    unsigned eff_n;                     // symbols that get coded
    unsigned largest;                   // largest of them
    unsigned distinct, triple_off;      // its (symbol, count, first position) triples in `triples`
    unsigned mode;                      // host: 0 = Huffman (dense table), 1 = Golomb
    unsigned m;                         // host: Golomb parameter
    unsigned reserved;
    unsigned long long bit_off;         // host: bit offset of the stream's first code in the container
    unsigned long long coded_bits;      // device: bits written for the stream's symbols
};

struct EntropyArgs {
    const uint16_t* counts;             // [n_lengths] the lengths stream
    const uint16_t* symbols;            // the 6K code streams back to back (StreamArgs::symbols)
    const unsigned long long* stream_off;   // [6K + 1]
    unsigned n_lengths;
    int n_streams;                      // 6K + 1
    uint16_t* packed;                   // run-length coded streams, at the offsets of their sources (capacity of `symbols`)
    EntStream* streams;                 // [n_streams]
    unsigned* totals;                   // [4]: scan blocks, (unused), triples written, triple overflow flag
    unsigned* blk_stream;               // per scan block: the stream it belongs to
    unsigned* blk_lead;                 // per scan block: symbols in front of its first run start | run ends there << 31
    unsigned* blk_inner;                // ... run-length symbols emitted for the runs that start inside the block
    unsigned* blk_tail;                 // ... symbols from its last run start to its end
    unsigned* blk_carry;                // ... symbols of the run its first symbol continues, in front of the block
    unsigned* blk_out;                  // ... first run-length symbol of the block in the packed stream
    unsigned* blk_bits;                 // ... code bits
    unsigned long long* blk_bit_off;    // ... bit offset of its first code in the container
    unsigned* ghist;                    // [n_streams][65536] zero between calls
    unsigned* gfirst;                   // [n_streams][65536] 0xFFFFFFFF between calls
    unsigned* triples;                  // [triple_cap][3], in (mapped) host memory: the compaction writes it across PCIe
    unsigned triple_cap;
    EntStream* host_streams;            // [n_streams] mirror of `streams` in (mapped) host memory
    unsigned* host_totals;              // [4] mirror of `totals`
    unsigned* tcode;                    // [n_streams][65536] zero between calls
    uint8_t* tlen;                      // [n_streams][65536]
    const unsigned* entries;            // [n_entries][3]: stream << 16 | symbol, code, length (may be mapped host memory)
    unsigned n_entries;
    unsigned* out32;                    // the container, zeroed
    unsigned long long out_words;
};
size_t entropy_max_blocks(unsigned long long symbols, int n_streams);
// capacity_symbols: upper bound of the symbols in all streams (lengths included); hipError_t as int
int launch_entropy_phase1(const EntropyArgs& a, unsigned long long capacity_symbols, void* stream);
// raw_symbols: symbols in all streams as assembled (sum of EntStream::n)
int launch_entropy_phase2(const EntropyArgs& a, unsigned long long raw_symbols, void* stream);

// choices[tiles][3][K] -> planar[3][K][tiles] (what the host entropy stage of mpc_encode_image(s) reads); hipError_t as int
int launch_planar_records(const uint32_t* choices, uint32_t* planar, long long tiles, int K, void* stream);

// bytes of workspace needed for `cap` tile-channels and K steps
size_t workspace_bytes(int cap, int K);
// carve a workspace out of one device allocation of workspace_bytes(cap, K) bytes
Workspace carve_workspace(void* device_mem, int cap, int K);

}  // namespace mpc
