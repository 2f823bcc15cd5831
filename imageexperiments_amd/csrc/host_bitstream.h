// host_bitstream.h -- product host code: the entropy stage and the ".mn" container of the
// CompressionLib codec.  It stays on the host (BASELINE.json north_star) and must emit the
// reference's bytes exactly:
//   bit buffer, zigzag, Golomb, Elias-Fano     CompressionLib/src/BitBuffer.cpp, inc/BitBuffer.h
//   canonical Huffman + u16 run-length code     CompressionLib/src/Huffman.cpp
//   Huffman-or-Golomb choice, container layout  CompressionLib/src/CompressedImage.cpp:359-460, 635-707
// Huffman ties are broken the way the reference's only toolchain (MSVC STL) breaks them; see
// MsvcHashOrder in the .cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace mpc {

class BitWriter {
public:
    void put(uint64_t value, int width);            // MSB first; width 0..64
    void append(const BitWriter& other);
    // put(code_of[s], length_of[s]) for every s in data[0..n), through a register accumulator (lengths 1..32)
    void put_codes(const uint16_t* data, size_t n, const uint32_t* code_of, const uint8_t* length_of, size_t total_bits);
    size_t bit_size() const { return nbits_; }
    const uint64_t* words() const { return words_.data(); }   // MSB-first, the last word zero-padded
    std::vector<uint8_t> bytes() const;             // zero-padded to a whole byte (BitBuffer::Save)
private:
    std::vector<uint64_t> words_;
    size_t nbits_ = 0;
};

class BitReader {
public:
    BitReader(const uint8_t* data, size_t nbytes) : p_(data), nbits_(8 * nbytes) {}
    // BitBuffer::ReadBits: the width is clipped to what remains; past the end it returns 0
    uint64_t get(int width);
    size_t remaining() const { return nbits_ - pos_; }
    // the next 32 bits left-aligned (zero beyond the end), without consuming them; skip() consumes
    uint32_t peek32() const;
    void skip(size_t bits) { pos_ += bits; }
    // for the decoders' fast loops (they read whole 64-bit words while at least eight bytes remain and fall back to get / peek32)
    const uint8_t* data() const { return p_; }
    size_t position() const { return pos_; }
    size_t size_bits() const { return nbits_; }
    void set_position(size_t bit) { pos_ = bit; }
private:
    const uint8_t* p_;
    size_t nbits_;
    size_t pos_ = 0;
};

inline uint32_t zigzag_encode(int32_t x) { return (static_cast<uint32_t>(x) << 1) ^ static_cast<uint32_t>(x >> 31); }
inline int32_t zigzag_decode(uint32_t x) { return static_cast<int32_t>((x >> 1) ^ static_cast<uint32_t>(-static_cast<int64_t>(x & 1))); }

uint32_t golomb_length(uint32_t value, uint32_t m);
void golomb_write(uint32_t value, uint32_t m, BitWriter& out);
uint32_t golomb_read(uint32_t m, BitReader& in);

// Elias-Fano code of a sorted sequence (BitBuffer.cpp:292-354): the Huffman symbol table uses it where it is shorter than raw
uint32_t elias_fano_length(size_t n, uint16_t max_symbol);
void elias_fano_write(const uint16_t* seq, size_t n, uint16_t max_symbol, BitWriter& out);
bool elias_fano_read(uint16_t* dst, size_t n, uint16_t max_symbol, BitReader& in);

void huffman_encode(const uint16_t* data, size_t n, BitWriter& out);
bool huffman_decode(BitReader& in, std::vector<uint16_t>& out);          // false = "Invalid bitstream"

std::vector<uint16_t> rle_encode(const uint16_t* data, size_t n);
std::vector<uint16_t> rle_decode(const uint16_t* data, size_t n);

void write_huffman_or_golomb(const uint16_t* data, size_t n, BitWriter& out);
bool read_huffman_or_golomb(BitReader& in, size_t length, std::vector<uint16_t>& out);

struct Streams {
    int width = 0, height = 0, K = 0, block_size = 0;
    uint16_t quant[3][32] = {};
    std::vector<uint16_t> lengths;                  // 3 per tile, x-outer / y-inner tile order
    std::vector<std::vector<uint16_t>> codes;       // [6K]: codes[2K*ch + 2i] deltaId, [+1] intCoeff of step i
};

// writeCompressed: codes are taken as the encoder holds them (DC coefficients not yet differenced)
std::vector<uint8_t> write_compressed(const Streams& s);
// readCompressed: codes come back with the DC differencing undone; false = invalid data
bool read_compressed(const uint8_t* bytes, size_t nbytes, Streams& out);

// Build the streams from per-tile records in the reference's visiting order (tile t = tx*tiles_y + ty):
// counts[t*3+ch], choices[(t*3+ch)*K + i] = deltaId | intCoeff << 16.   (encodeImage, CompressedImage.cpp:555-572)
Streams assemble_streams(int width, int height, int K, int block_size, const double* quant /*[3*K]*/,
                         const uint16_t* counts, const uint32_t* choices);

// assemble_streams + write_compressed without the intermediate streams (what mpc_assemble_streams / mpc_encode_image use)
std::vector<uint8_t> encode_records(int width, int height, int K, int block_size, const double* quant /*[3*K]*/,
                                    const uint16_t* counts, const uint32_t* choices);
// the same into a malloc'ed buffer (release with free); nullptr = out of memory
uint8_t* encode_records_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                               const uint32_t* choices, size_t* nbytes);
// same container from records in planar order, planar[(ch * K + i) * tiles + t]: every (channel, step) job then reads
// one contiguous run instead of one word per 12*K bytes (what mpc_encode_image downloads after the device transposes)
uint8_t* encode_planar_records_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                      const uint32_t* planar, size_t* nbytes);

// same container from streams assembled on the device (mp_streams.hip): symbols = codes[0] ++ codes[1] ++ ... (live symbols
// only, step-0 coefficients already difference coded), off[6K + 1] = stream boundaries in symbols
uint8_t* encode_symbol_streams_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                      const uint16_t* symbols, const unsigned long long* off, size_t* nbytes);

// ---- the host's share of the entropy stage when the device does the per-symbol work (mp_entropy.hip) ----
struct StreamPlan {
    BitWriter pre;                   // what precedes the stream's codes: run-length flag (+ size), Huffman/Golomb bit, table or M
    BitWriter post;                  // what follows them: the pseudo-EOF code (Huffman)
    size_t payload_bits = 0;         // bits of the codes themselves
    int mode = 0;                    // 0 = Huffman, 1 = Golomb
    uint32_t m = 0;                  // Golomb parameter
    int max_code_length = 0;         // Huffman
    std::vector<uint32_t> entries;   // Huffman: (symbol, code, length) of every symbol that occurs
};
// One stream of the container from its statistics: n coded symbols (after run-length coding if `shorter`), `largest` of them,
// triples[3 * distinct] = (symbol, count, position of first appearance) in any order.  rle_flag: false for `lengths`.
void plan_stream(bool rle_flag, bool shorter, uint32_t rle_size, size_t n, uint32_t largest, const uint32_t* triples, size_t distinct,
                 StreamPlan& plan);
BitWriter container_head(int width, int height, int K, int block_size, const double* quant /*[3*K]*/);
// dst |= piece, MSB first, at bit_offset (dst: the container's bytes, dst_bytes of them writable)
void or_bits(uint8_t* dst, size_t dst_bytes, size_t bit_offset, const BitWriter& piece);
// body(0..n-1) on the entropy stage's worker pool
void parallel_jobs(int n, const std::function<void(int)>& body);
// the same on a second pool of at most `workers` threads (the calling thread included): copies that run beside the entropy stage
void parallel_io_jobs(int n, int workers, const std::function<void(int)>& body);

// encode_symbol_streams_malloc through plan_stream / or_bits, the device's share (statistics, code writing) done on the host
uint8_t* encode_symbol_streams_by_plan_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                              const uint16_t* symbols, const unsigned long long* off, size_t* nbytes);

// Inverse of assemble_streams: per-tile records in the reference's visiting order.  counts[3*tiles],
// choices[3*tiles*K] (deltaId | intCoeff << 16, zero beyond count).  false = streams inconsistent with `lengths`.
bool disassemble_streams(const Streams& s, std::vector<uint16_t>& counts, std::vector<uint32_t>& choices);
// same into caller-provided storage (3*tiles and 3*tiles*K elements); entries beyond a count are left untouched
bool disassemble_streams(const Streams& s, uint16_t* counts, uint32_t* choices);

}  // namespace mpc
