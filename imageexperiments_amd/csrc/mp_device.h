// mp_device.h -- product: device-side parameter block and launchers for the
// MI355X (gfx950) quantized matching-pursuit tile encoder.
#pragma once
#include <cstddef>
#include <cstdint>

namespace mpc {

// One launch = one stripe of tile rows of one RGB frame.
struct EncodeParams {
    // input frame (device memory), row-major, 3 B/pixel (img::image<rgb>, image.h:123-131)
    const uint8_t* rgb;
    int width, height;
    long long row_stride;            // bytes between image rows
    int frames;                      // batch of equally sized frames, `frame_stride` bytes apart
    long long frame_stride;
    // stripe of tile rows handled by this launch
    int tile_row_begin, tile_rows;   // rows [begin, begin+tile_rows)
    int tiles_x;
    int K;                           // MP steps (<= 32)
    // dictionary (device memory)
    const double* base;              // [base_rows_padded][64] row-major (rows >= num_base are zero)
    int num_base;                    // 510
    int base_rows_padded;            // multiple of 2
    const double* detail;            // [3][detail_rows][64] row-major
    const double* detail_t;          // [3][num_base][32][64][2] transposed + padded blocks
    long long detail_rows;           // rows per channel (31 622)
    const int32_t* block_rows;       // [num_base]
    const int32_t* block_row_off;    // [num_base+1]
    const double* quant;             // [3][K] (device memory)
    // outputs (device memory); tile index t = frame*tiles + tx*tile_rows + (ty - tile_row_begin)
    uint16_t* counts;                // [tiles][3]
    uint32_t* choices;               // [tiles][3][K]  lo16 = deltaId, hi16 = intCoeff (BasisChoice layout)
    double* energy;                  // [tiles][3]  sum of squared residual at termination
    uint32_t* swept;                 // [tiles][3]  dictionary rows correlated (SURVEY 8d "S")
    // work queues: queue[0] hands out luma (Y) tile-channels, queue[1] chroma (U,V) ones; both zeroed before
    // the launch.  Lanes < y_lanes of every wave prefer the Y queue, the others the chroma queue; a lane whose
    // preferred queue is dry takes from the other one.  (Y pursuits run ~4x longer than chroma ones: starting
    // all of them at once, spread over every wave, and streaming chroma through the remaining lanes is the
    // longest-job-first schedule.)
    unsigned int* queue;
    int y_lanes;
    // optional profiling: if non-null, per-phase shader-clock totals [refill, base sweep, detail sweep, finish, iterations]
    unsigned long long* phase_cycles;
    // vector mode (matching::CalcMPDynamic on caller-supplied 64-vectors instead of image tiles):
    // when vec_in != nullptr the tasks are vec_count vectors of channel vec_channel, out index = vector index
    const double* vec_in;            // [vec_count][64]
    int vec_count;
    int vec_channel;
};

struct HistParams {
    const uint16_t* counts;          // [tiles][3]
    const uint32_t* choices;         // [tiles][3][K]
    long long tiles;
    int K;
    uint32_t* hist;                  // [(1 + 6K)][8192]
};

constexpr int kHistBins = 8192;
constexpr int kMaxDeviceK = 32;

// Launch the encoder on `stream` (hipStream_t as void*). `waves` = grid size (one wave64 per workgroup).
// Returns hipError_t as int.
int launch_encode(const EncodeParams& p, int waves, void* stream);
int launch_histogram(const HistParams& p, void* stream);

// resident-wave capacity for the encode kernel on the current device
int encode_max_resident_waves();

}  // namespace mpc
