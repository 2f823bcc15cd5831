// host_dictionary.h -- product host code (C++), MI355X tile codec.
//
// Builds, once per context, everything the device needs for the
// CompressionLib tile-encode path and lays it out for HBM:
//   * the segmented line-cut base dictionary      (reference: BasisSet.cpp:204-380)
//   * the per-segment KLT detail blocks, 3 channels (reference: BasisSet.cpp:513-616,
//     createBasis :118-152, SymmetricEigenDecomposition symmeigen.cpp:34-244)
//   * the bit-allocation quantisation tables       (reference: CompressedImage.cpp:124-166)
// It replaces compressed::createCompressionContext (CompressedImage.cpp:291-315).
// All arithmetic is IEEE double in the reference's operation order
// (compiled with -ffp-contract=off) so that the tables are bit-identical to
// the reference's double path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mpc {

constexpr int kMaxK = 32;
constexpr int kTileN = 64;        // 8x8 tiles: the only block size the device path implements

struct LineCut { int ax, ay, bx, by; };

struct Dictionary {
    int block_size = 8;
    int n = kTileN;                       // pixels per tile
    int num_base = 0;                     // 510 for 8x8
    std::vector<LineCut> cuts;            // [num_base]
    std::vector<double> base;             // [num_base][n]            row-major
    std::vector<int32_t> block_rows;      // [num_base]   rows of DetailBasis[i] (62 | 63)
    std::vector<int32_t> block_row_off;   // [num_base+1] prefix sum of block_rows
    std::vector<double> detail[3];        // [block_row_off.back()][n] row-major, per channel
    int total_detail_rows() const { return block_row_off.empty() ? 0 : block_row_off.back(); }
};

// Symmetric eigen-decomposition; vectors returned as columns of `vec` (row-major n*n).
void symmetric_eigen(const double* a, int n, double* vec, double* val);

// KLT basis of a covariance matrix: rows sorted by (|lambda|, index) descending,
// first significant entry made positive.
void klt_basis(const double* cov, int n, double* rows);

double covariance_model(int channel, double dx, double dy);

std::vector<LineCut> distinct_line_cuts(int block_size);

Dictionary build_dictionary(int block_size);

// quant[ch*K + i]
void quantisation_tables(int K, int block_size, double bpp, double* quant);

// Device-oriented re-layouts -------------------------------------------------
// base padded to a multiple of `pad_rows` rows with zero rows (a zero row can never be selected:
// Select starts from bestCoeff = 0 with a strict '>', MatchingPursuit.cpp:9-19).
std::vector<double> base_padded(const Dictionary& d, int pad_rows, int* padded_rows);

// Reduced-precision copy of `nrows` dictionary rows (row-major doubles, n = 64) for the device's filter pass: every
// element x is split into two bfloat16 values hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to 2^-16 relative), the rows
// are padded with zero rows to `tiles` tiles of 16 and laid out in the operand order of v_mfma_f32_16x16x32_bf16:
//   out[((tile*4 + 2*kk + part)*64 + lane)*8 + j] = part(rows[tile*16 + (lane & 15)][32*kk + 8*(lane >> 4) + j])
// (part 0 = hi, 1 = lo; kk = 0, 1), so that one 16-byte load per lane is one MFMA operand.  4 KiB per tile.
// A row that is bit-for-bit +-(an earlier row) is left zero: its projection ties with the earlier row's exactly
// (IEEE negation commutes with every rounding of the dot product), so Select()'s strict '>' never returns it.
// k_order selects which pixel an operand element holds (the MFMA sums over k in any order as long as both operands
// agree): 0 = 32*kk + 8*(lane >> 4) + j (the step-synchronous filter kernels), 1 = 16*pos(lane >> 4) + 8*kk + j with pos = 0, 1, 3, 2
// (the persistent pursuit kernel: a lane's 16 elements are 16 CONSECUTIVE pixels, which is what its exact evaluation wants;
// the order of the lane rows is the order in which one-instruction lane swaps pass the running sum on).
// shadow_out (optional): 1 for every row left zero because it duplicates an earlier one.
std::vector<uint16_t> filter_tiles(const double* rows, int nrows, int tiles, int k_order = 0,
                                   std::vector<uint8_t>* shadow_out = nullptr);
constexpr int kFilterTileHalves = 2048;       // 16-bit elements per tile
uint16_t bf16_round(float x);                 // round to nearest even

}  // namespace mpc
