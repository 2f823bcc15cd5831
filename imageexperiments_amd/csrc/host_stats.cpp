// host_stats.cpp -- product host code (see host_stats.h).  Build with -ffp-contract=off.
#include "host_stats.h"

#include <charconv>
#include <cmath>

namespace mpc {

void RunningStat::update(double val) {
    if (N == 0) {
        N = 1.0;
        min = max = mean = val;
        sumSq = 0.0;
        return;
    }
    if (val < min) min = val;
    if (val > max) max = val;
    N += 1.0;
    const double before = val - mean;
    mean += before / N;
    const double after = val - mean;
    sumSq += before * after;
}

PatchStats::PatchStats(int K_, int block_size_, uint32_t seed) : K(K_), block_size(block_size_) {
    rand.seed(seed);
    for (int ch = 0; ch < 3; ++ch) {
        coeff[ch].assign(static_cast<size_t>(K), RunningStat{});
        select[ch].assign(static_cast<size_t>(K), RunningStat{});
    }
}

void PatchStats::sample_origins(int width, int height, int patches, std::vector<int>& xs, std::vector<int>& ys) {
    xs.resize(static_cast<size_t>(patches));
    ys.resize(static_cast<size_t>(patches));
    const uint32_t span_x = static_cast<uint32_t>(width - block_size), span_y = static_cast<uint32_t>(height - block_size);
    for (int p = 0; p < patches; ++p) {
        xs[p] = static_cast<int>(static_cast<uint32_t>(rand()) % span_x);
        ys[p] = static_cast<int>(static_cast<uint32_t>(rand()) % span_y);
    }
}

void PatchStats::accumulate(const uint16_t* counts, const uint32_t* choices, int patches) {
    for (int p = 0; p < patches; ++p)
        for (int ch = 0; ch < 3; ++ch) {
            const int count = counts[p * 3 + ch];
            const uint32_t* rec = choices + (static_cast<size_t>(p) * 3 + ch) * K;
            for (int i = 0; i < count && i < K; ++i) {
                coeff[ch][i].update(static_cast<double>(rec[i] >> 16));          // intCoeff (zigzag code)
                select[ch][i].update(static_cast<double>(rec[i] & 0xffffu));     // deltaId
            }
        }
}

std::string format_double(double v) {
    char buf[64];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v);
    return std::string(buf, r.ptr);
}

std::string PatchStats::report() const {
    static const char* const names[3] = {"Y", "U", "V"};
    std::string out;
    auto line = [&](const char* what, int i, const RunningStat& s) {
        const double var = s.sample_variance();
        out += what;
        out += ' ';
        out += std::to_string(i);
        out += " min " + format_double(s.min) + " max " + format_double(s.max) + " range " + format_double(s.max - s.min);
        out += " mean " + format_double(s.mean) + " variance " + format_double(var) + " std dev " + format_double(std::sqrt(var));
        out += '\n';
    };
    for (int ch = 0; ch < 3; ++ch) {
        out += std::string(names[ch]) + " coeff stats\n";
        for (int i = 0; i < K; ++i) line("coeff", i, coeff[ch][i]);
        out += std::string(names[ch]) + " basisId stats\n";
        for (int i = 0; i < K; ++i) line("deltaId", i, select[ch][i]);
    }
    return out;
}

}  // namespace mpc
