// host_codec.cpp -- product host code (see host_codec.h).  Build with -ffp-contract=off.
#include "host_codec.h"

#include <algorithm>
#include <cmath>

namespace mpc {

namespace {

inline uint8_t to_u8(double v) {                       // static_cast<uchar>(std::clamp(std::round(v), 0.0, 255.0))
    return static_cast<uint8_t>(std::clamp(std::round(v), 0.0, 255.0));
}

// FromCoeffsDynamic: sum of coeff * dictionary row over the recorded steps; the dynamic dictionary is walked
// through a row table (base rows, then DetailBasis[choice] of every earlier base choice, repeats included).
bool reconstruct(const Dictionary& d, int ch, const uint16_t* quant, int count, const uint16_t* delta, const uint16_t* coef,
                 double* out, std::vector<const double*>& rows) {
    const int n = d.n;
    rows.clear();
    for (int a = 0; a < d.num_base; ++a) rows.push_back(d.base.data() + static_cast<size_t>(a) * n);
    int choice = 0;
    for (int i = 0; i < count; ++i) {
        choice = (i > 0) ? choice + zigzag_decode(delta[i]) : static_cast<int>(delta[0]);
        if (choice >= 0 && choice < d.num_base) {
            const double* blk = d.detail[ch].data() + static_cast<size_t>(d.block_row_off[choice]) * n;
            for (int a = 0; a < d.block_rows[choice]; ++a) rows.push_back(blk + static_cast<size_t>(a) * n);
        }
    }
    std::fill(out, out + n, 0.0);
    choice = 0;
    for (int i = 0; i < count; ++i) {
        choice = (i > 0) ? choice + zigzag_decode(delta[i]) : static_cast<int>(delta[0]);
        if (choice < 0 || choice >= static_cast<int>(rows.size())) return false;   // the reference's bounds-checked [] throws
        const double c = static_cast<double>(quant[i]) * static_cast<double>(zigzag_decode(coef[i]));
        const double* basis = rows[static_cast<size_t>(choice)];
        for (int j = 0; j < n; ++j) out[j] += basis[j] * c;
    }
    return true;
}

}  // namespace

bool decode_streams(const Dictionary& dict, const Streams& s, std::vector<uint8_t>& rgb) {
    const int K = s.K, bs = s.block_size, W = s.width, H = s.height, n = bs * bs;
    if (bs != dict.block_size) return false;
    rgb.assign(static_cast<size_t>(W) * H * 3, 0);
    std::vector<size_t> cursor(static_cast<size_t>(3 * K), 0);
    std::vector<const double*> rows;
    std::vector<double> plane(static_cast<size_t>(3 * n));
    uint16_t delta[32], coef[32];
    size_t at = 0;
    for (int x = 0; x < W; x += bs)
        for (int y = 0; y < H; y += bs) {
            for (int ch = 0; ch < 3; ++ch) {
                if (at >= s.lengths.size()) return false;
                const int count = s.lengths[at++];
                if (count > K) return false;
                for (int i = 0; i < count; ++i) {
                    const std::vector<uint16_t>& cd = s.codes[2 * K * ch + 2 * i];
                    const std::vector<uint16_t>& cc = s.codes[2 * K * ch + 2 * i + 1];
                    size_t& c = cursor[static_cast<size_t>(K * ch + i)];
                    if (c >= cd.size() || c >= cc.size()) return false;
                    delta[i] = cd[c];
                    coef[i] = cc[c++];
                }
                if (!reconstruct(dict, ch, s.quant[ch], count, delta, coef, plane.data() + static_cast<size_t>(ch) * n, rows))
                    return false;
            }
            for (int dx = 0; dx < bs; ++dx)
                for (int dy = 0; dy < bs; ++dy) {
                    const int u = x + dx, v = y + dy;
                    if (u >= W || v >= H) continue;
                    const double Y = plane[dx + bs * dy], U = plane[n + dx + bs * dy], V = plane[2 * n + dx + bs * dy];
                    uint8_t* px = rgb.data() + 3 * (static_cast<size_t>(v) * W + u);
                    px[0] = to_u8(Y + 1.13983 * V);
                    px[1] = to_u8(Y - 0.39466 * U - 0.58060 * V);
                    px[2] = to_u8(Y + 2.03211 * U);
                }
        }
    return true;
}

double psnr(const uint8_t* a, const uint8_t* b, int W, int H) {
    double se = 0.0;
    for (int x = 0; x < W; ++x)                         // x outer, y inner like the reference (summation order)
        for (int y = 0; y < H; ++y) {
            const uint8_t* pa = a + 3 * (static_cast<size_t>(y) * W + x);
            const uint8_t* pb = b + 3 * (static_cast<size_t>(y) * W + x);
            const double r = static_cast<double>(pb[0]) - static_cast<double>(pa[0]);
            const double g = static_cast<double>(pb[1]) - static_cast<double>(pa[1]);
            const double bl = static_cast<double>(pb[2]) - static_cast<double>(pa[2]);
            se += r * r + g * g + bl * bl;
        }
    const double mse = se / static_cast<double>(static_cast<size_t>(W) * static_cast<size_t>(H));
    return 20.0 * std::log10(3.0 * 255.0) - 10.0 * std::log10(mse);
}

}  // namespace mpc
