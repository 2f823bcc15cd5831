// host_codec.cpp -- product host code (see host_codec.h).  Build with -ffp-contract=off.
#include "host_codec.h"

#include <cmath>

namespace mpc {

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
