// host_stats.h -- product host code: the "-s" patch statistics of Compression.cpp:200-302.
//   RunningStat       math::Stat (update / sampleVariance)     SimpleMatrix/src/covariance.cpp:5-25
//   PatchStats        the six Stat[K] arrays, the std::mt19937 patch sampler and the text report
// The pursuit itself runs on the device: the sampled patches are laid out as the tiles of a one-tile-high mosaic
// and go through the ordinary tile encoder with all quantisers 1.0 (Compression.cpp:221-225).
#pragma once
#include <cstdint>
#include <random>
#include <string>
#include <vector>

namespace mpc {

struct RunningStat {
    double N = 0.0, min = 0.0, max = 0.0, mean = 0.0, sumSq = 0.0;
    void update(double val);
    double sample_variance() const { return sumSq / (N - 1.0); }
};

struct PatchStats {
    int K = 0, block_size = 8;
    std::mt19937 rand;
    std::vector<RunningStat> coeff[3], select[3];           // [channel][step]: intCoeff and deltaId statistics

    PatchStats(int K_, int block_size_, uint32_t seed);
    // patch origins in the reference's order: x = rand() % (width - bs), y = rand() % (height - bs)
    void sample_origins(int width, int height, int patches, std::vector<int>& xs, std::vector<int>& ys);
    // records of `patches` patches as the tile encoder returns them: counts[p*3+ch], choices[(p*3+ch)*K + i]
    void accumulate(const uint16_t* counts, const uint32_t* choices, int patches);
    // the file Compression.cpp:275-301 writes ("{}" of std::format = shortest round-trip text)
    std::string report() const;
};

// shortest round-trip text of a double as std::format("{}", v) prints it
std::string format_double(double v);

}  // namespace mpc
