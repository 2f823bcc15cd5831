// host_dictionary.cpp -- product host code: dictionary + quantisation tables.
// See host_dictionary.h for the reference citations.  Build with -ffp-contract=off.
#include "host_dictionary.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <numeric>
#include <stdexcept>
#include <utility>

namespace mpc {

namespace {

// ---------------------------------------------------------------------------
// Symmetric eigensolver (reference: SimpleMatrix/src/symmeigen.cpp:34-244).
// Householder reduction from the last row upwards, accumulation of the
// reflectors, then implicit-shift QL sweeps.  The operation order inside each
// statement is the reference's; only the packaging differs.
// ---------------------------------------------------------------------------
class SymEig {
public:
    SymEig(const double* a, int n) : n_(n), z_(a, a + static_cast<size_t>(n) * n), d_(n, 0.0), e_(n, 0.0) {}

    void run() {
        householder();
        form_q();
        ql();
    }
    const std::vector<double>& vectors() const { return z_; }
    const std::vector<double>& values() const { return d_; }

private:
    // the reference's abs(): x > 0 ? x : -x   (symmeigen.cpp:9-10)
    static double mag(double x) { return x > 0 ? x : -x; }

    static double pythag(double a, double b) {          // symmeigen.cpp:18-32
        if (mag(a) > mag(b)) {
            const double q = b / a;
            return mag(a) * std::sqrt(1.0 + q * q);
        }
        if (b != 0.0) {
            const double q = a / b;
            return mag(b) * std::sqrt(1.0 + q * q);
        }
        return 0.0;
    }

    double& z(int r, int c) { return z_[static_cast<size_t>(r) * n_ + c]; }

    void householder() {                                 // symmeigen.cpp:47-124
        const int n = n_;
        for (int c = 0; c < n; ++c) d_[c] = z(n - 1, c);
        for (int i = n - 1; i > 0; --i) {
            double scale = 0.0;
            double h = 0.0;
            for (int j = 0; j < i; ++j) scale = scale + mag(d_[j]);
            if (scale == 0.0) {
                e_[i] = d_[i - 1];
                for (int j = 0; j < i; ++j) {
                    d_[j] = z(i - 1, j);
                    z(i, j) = 0.0;
                    z(j, i) = 0.0;
                }
            } else {
                for (int j = 0; j < i; ++j) {
                    const double s = d_[j] / scale;
                    d_[j] = s;
                    h += s * s;
                }
                double f = d_[i - 1];
                double g = std::sqrt(h);
                if (f > 0) g = -g;
                e_[i] = scale * g;
                h = h - f * g;
                d_[i - 1] = f - g;
                std::fill(e_.begin(), e_.begin() + i, 0.0);
                for (int j = 0; j < i; ++j) {
                    f = d_[j];
                    z(j, i) = f;
                    g = e_[j] + z(j, j) * f;
                    for (int k = j + 1; k <= i - 1; ++k) {
                        g += z(k, j) * d_[k];
                        e_[k] += z(k, j) * f;
                    }
                    e_[j] = g;
                }
                f = 0.0;
                for (int j = 0; j < i; ++j) {
                    e_[j] /= h;
                    f += e_[j] * d_[j];
                }
                const double hh = f / (h + h);
                for (int j = 0; j < i; ++j) e_[j] -= hh * d_[j];
                for (int j = 0; j < i; ++j) {
                    f = d_[j];
                    g = e_[j];
                    for (int k = j; k <= i - 1; ++k) z(k, j) -= (f * e_[k] + g * d_[k]);
                    d_[j] = z(i - 1, j);
                    z(i, j) = 0.0;
                }
            }
            d_[i] = h;
        }
    }

    void form_q() {                                      // symmeigen.cpp:126-160
        const int n = n_;
        for (int i = 0; i < n - 1; ++i) {
            z(n - 1, i) = z(i, i);
            z(i, i) = 1.0;
            const double h = d_[i + 1];
            if (h != 0.0) {
                for (int j = 0; j <= i; ++j) d_[j] = z(j, i + 1) / h;
                for (int j = 0; j <= i; ++j) {
                    double g = 0.0;
                    for (int k = 0; k <= i; ++k) g += z(k, i + 1) * z(k, j);
                    for (int k = 0; k <= i; ++k) z(k, j) -= g * d_[k];
                }
            }
            for (int j = 0; j <= i; ++j) z(j, i + 1) = 0.0;
        }
        for (int i = 0; i < n; ++i) {
            d_[i] = z(n - 1, i);
            z(n - 1, i) = 0.0;
        }
        z(n - 1, n - 1) = 1.0;
        e_[0] = 0.0;
    }

    void ql() {                                          // symmeigen.cpp:162-243
        const int n = n_;
        const double eps = 1.0E-20;
        double shift = 0.0;
        double pivot = 0.0;
        for (int i = 1; i < n; ++i) e_[i - 1] = e_[i];
        e_[n - 1] = 0.0;
        for (int i = 0; i < n; ++i) {
            pivot = std::max(pivot, mag(d_[i]) + mag(e_[i]));
            int l = i;
            for (; l < n; ++l)
                if (mag(e_[l]) <= eps * pivot) break;
            if (l > i) {
                do {
                    double f = d_[i];
                    double g = (d_[i + 1] - f) / (2.0 * e_[i]);
                    double r = pythag(g, 1.0);
                    if (g < 0) r = -r;
                    d_[i] = e_[i] / (g + r);
                    d_[i + 1] = e_[i] * (g + r);
                    const double d_next = d_[i + 1];
                    double h = f - d_[i];
                    for (int j = i + 2; j < n; ++j) d_[j] -= h;
                    shift = shift + h;
                    g = d_[l];
                    const double e_next = e_[i + 1];
                    double c = 1.0, c_prev = 1.0, c_prev2 = 1.0;
                    double s = 0.0, s_prev = 0.0;
                    for (int j = l - 1; j >= i; --j) {
                        c_prev2 = c_prev;
                        c_prev = c;
                        s_prev = s;
                        f = c * e_[j];
                        h = c * g;
                        r = pythag(g, e_[j]);
                        e_[j + 1] = s * r;
                        s = e_[j] / r;
                        c = g / r;
                        g = c * d_[j] - s * f;
                        d_[j + 1] = h + s * (c * f + s * d_[j]);
                        for (int k = 0; k < n; ++k) {
                            h = z(k, j + 1);
                            z(k, j + 1) = s * z(k, j) + c * h;
                            z(k, j) = c * z(k, j) - s * h;
                        }
                    }
                    g = -s * s_prev * c_prev2 * e_next * e_[i] / d_next;
                    e_[i] = s * g;
                    d_[i] = c * g;
                } while (mag(e_[i]) > eps * pivot);
            }
            d_[i] = d_[i] + shift;
            e_[i] = 0.0;
        }
    }

    int n_;
    std::vector<double> z_, d_, e_;
};

// Signed distance test of BasisSet.cpp:188-190: integer cross product over the
// square root of an integer.  A zero-length line yields 0/0 = NaN, for which
// ">= 0" is false -- kept on purpose, it decides which masks are generated.
inline bool on_positive_side(const LineCut& l, int px, int py) {
    const int cross = (l.bx - l.ax) * (l.ay - py) - (l.ax - px) * (l.by - l.ay);
    const int len2 = (l.bx - l.ax) * (l.bx - l.ax) + (l.by - l.ay) * (l.by - l.ay);
    return static_cast<double>(cross) / std::sqrt(static_cast<double>(len2)) >= 0.0;
}

// Pixel mask of a cut, bit (x + y*bs).
uint64_t cut_mask(const LineCut& l, int bs) {
    uint64_t m = 0;
    for (int x = 0; x < bs; ++x)
        for (int y = 0; y < bs; ++y)
            if (on_positive_side(l, x, y)) m |= 1ULL << (x + y * bs);
    return m;
}

// std::map ordering of the reference (ShapeComparator, BasisSet.cpp:192-202):
// compare element by element from index 0; the vector holding `true` at the
// first difference sorts first.
struct MaskOrder {
    bool operator()(uint64_t a, uint64_t b) const {
        const uint64_t diff = a ^ b;
        if (diff == 0) return false;
        const uint64_t lowest = diff & (~diff + 1);
        return (a & lowest) != 0;
    }
};

}  // namespace

void symmetric_eigen(const double* a, int n, double* vec, double* val) {
    if (n <= 0) return;
    SymEig s(a, n);
    s.run();
    std::memcpy(vec, s.vectors().data(), sizeof(double) * static_cast<size_t>(n) * n);
    std::memcpy(val, s.values().data(), sizeof(double) * static_cast<size_t>(n));
}

// BasisSet.cpp:118-152
void klt_basis(const double* cov, int n, double* rows) {
    if (n <= 0) return;
    SymEig s(cov, n);
    s.run();
    const std::vector<double>& vec = s.vectors();
    const std::vector<double>& val = s.values();
    std::vector<std::pair<double, int>> order;
    order.reserve(n);
    for (int i = 0; i < n; ++i) order.emplace_back(std::fabs(val[i]), i);
    std::sort(order.begin(), order.end(), std::greater<std::pair<double, int>>());
    for (int i = 0; i < n; ++i) {
        double* out = rows + static_cast<size_t>(i) * n;
        const int col = order[i].second;
        for (int k = 0; k < n; ++k) out[k] = vec[static_cast<size_t>(k) * n + col];
        const double* first = std::find_if(out, out + n, [](double v) { return std::fabs(v) > 1E-10; });
        if (first != out + n && *first < 0.0)
            for (int k = 0; k < n; ++k) out[k] = -out[k];
    }
}

// BasisSet.cpp:12-24
double covariance_model(int channel, double dx, double dy) {
    if (channel == 0)
        return 3817.7299999999996 * std::exp(-1.48854e-05 * dx * dx + -1.7273e-05 * dy * dy) +
               657.8100000000001 * std::exp(-0.0436057 * std::fabs(dx) + -0.050844400000000005 * std::fabs(dy));
    if (channel == 1) return 241.49 * std::exp(-0.00134755 * std::fabs(dx) + -0.00147572 * std::fabs(dy));
    return 371.87199999999996 * std::exp(-0.00147084 * std::fabs(dx) + -0.0015265799999999998 * std::fabs(dy));
}

// BasisSet.cpp:204-297
std::vector<LineCut> distinct_line_cuts(int bs) {
    if (bs < 1 || bs * bs > 64) throw std::invalid_argument("block size must satisfy 1 <= bs*bs <= 64");
    const uint64_t full = (bs * bs == 64) ? ~0ULL : ((1ULL << (bs * bs)) - 1ULL);
    std::map<uint64_t, LineCut, MaskOrder> shapes;
    auto add_if_new = [&](const LineCut& l) {               // :219-224 (mask only)
        const uint64_t m = cut_mask(l, bs);
        if (!shapes.count(m)) shapes.emplace(m, l);
    };
    auto add_if_new_either = [&](const LineCut& l) {        // :249-257 (mask or its complement)
        const uint64_t m = cut_mask(l, bs);
        if (!shapes.count(m) && !shapes.count(~m & full)) shapes.emplace(m, l);
    };
    for (int side = -1; side < bs + 1; ++side) {
        add_if_new(LineCut{0, side, bs, side});
        add_if_new(LineCut{side, 0, side, bs});
    }
    for (int p = -bs; p < 2 * bs; ++p)
        for (int q = -bs; q < 2 * bs; ++q) {
            add_if_new_either(LineCut{p, -bs, -bs, q});
            add_if_new_either(LineCut{p, -bs, q, bs});
            add_if_new_either(LineCut{p, -bs, bs, q});
        }
    for (int p = -bs; p < 2 * bs; ++p)
        for (int q = -bs; q < 2 * bs; ++q) {
            add_if_new_either(LineCut{bs, p, -bs, q});
            add_if_new_either(LineCut{bs, p, q, bs});      // the reference's 3rd candidate repeats this one (:279,:289)
        }
    std::vector<LineCut> out;
    out.reserve(shapes.size());
    for (const auto& kv : shapes) out.push_back(kv.second);
    return out;
}

namespace {

// BasisSet.cpp:299-380: +-1 mask on a 2x supersampled grid, 7x7 Gaussian (sigma 1)
// with clamped borders, mean removed unless the mask is constant, unit norm.
void segment_atoms(int bs, const std::vector<LineCut>& cuts, std::vector<double>& out) {
    const int n = bs * bs;
    const int fine = 2 * bs;
    const double sigma = 1.0;
    const int taps = static_cast<int>(1 + sigma * 6);
    const int half = taps / 2;
    std::vector<double> kernel(static_cast<size_t>(taps) * taps);
    for (int dx = -half; dx <= half; ++dx)
        for (int dy = -half; dy <= half; ++dy)
            kernel[(dx + half) + taps * (dy + half)] =
                std::exp(-static_cast<double>(dx * dx + dy * dy) / (2.0 * sigma * sigma));
    out.assign(cuts.size() * static_cast<size_t>(n), 0.0);
    std::vector<double> sign(static_cast<size_t>(fine) * fine);
    for (size_t a = 0; a < cuts.size(); ++a) {
        double* atom = out.data() + a * n;
        const LineCut fine_cut{cuts[a].ax * 2, cuts[a].ay * 2, cuts[a].bx * 2, cuts[a].by * 2};
        bool any_pos = false, any_neg = false;
        for (int x = 0; x < fine; ++x)
            for (int y = 0; y < fine; ++y) {
                const bool pos = on_positive_side(fine_cut, x, y);
                (pos ? any_pos : any_neg) = true;
                sign[x + fine * y] = pos ? +1.0 : -1.0;
            }
        const bool constant = !(any_pos && any_neg);
        double total = 0.0;
        for (int x = 0; x < bs; ++x)
            for (int y = 0; y < bs; ++y) {
                double weight = 0.0, acc = 0.0;
                for (int dx = -half; dx <= half; ++dx) {
                    const int u = std::clamp(2 * x + dx, 0, fine - 1);
                    for (int dy = -half; dy <= half; ++dy) {
                        const int v = std::clamp(2 * y + dy, 0, fine - 1);
                        const double w = kernel[(dx + half) + taps * (dy + half)];
                        weight += w;
                        acc += w * sign[u + fine * v];
                    }
                }
                const double value = acc / weight;
                total += value;
                atom[x + bs * y] = value;
            }
        const double mean = total / static_cast<double>(n);
        double sumsq = 0.0;
        for (int j = 0; j < n; ++j) {
            double value = atom[j];
            if (!constant) {
                value -= mean;
                atom[j] = value;
            }
            sumsq += value * value;
        }
        const double norm = std::sqrt(sumsq);
        if (sumsq != 0.0)
            for (int j = 0; j < n; ++j) atom[j] = atom[j] / norm;
    }
}

struct Side {
    std::vector<int> pos;          // pixel index x + bs*y, in the reference's x-outer/y-inner visiting order
    std::vector<int> xs, ys;
    std::vector<double> klt;       // [count][count]
};

// one detail row (BasisSet.cpp:561-584 / :587-612)
void spread_row(const Side& s, int vec_index, bool guard_zero_norm, int n, double* dst) {
    const int cnt = static_cast<int>(s.pos.size());
    const double* v = s.klt.data() + static_cast<size_t>(vec_index) * cnt;
    std::fill(dst, dst + n, 0.0);
    double mean = 0.0;
    for (int j = 0; j < cnt; ++j) {
        mean += v[j];
        dst[s.pos[j]] = v[j];
    }
    mean /= static_cast<double>(cnt);
    double ss = 0.0;
    for (int j = 0; j < cnt; ++j) {
        double value = dst[s.pos[j]];
        value -= mean;
        ss += value * value;
        dst[s.pos[j]] = value;
    }
    ss = std::sqrt(ss);
    const bool divide = !guard_zero_norm || ss != 0.0;
    for (int j = 0; j < n; ++j)
        if (divide) dst[j] = dst[j] / ss;
}

// BasisSet.cpp:513-616: returns the rows appended
int detail_block(int bs, const LineCut& cut, int channel, std::vector<double>& out) {
    const int n = bs * bs;
    Side side[2];
    for (int x = 0; x < bs; ++x)
        for (int y = 0; y < bs; ++y) {
            Side& s = side[on_positive_side(cut, x, y) ? 0 : 1];
            s.pos.push_back(x + bs * y);
            s.xs.push_back(x);
            s.ys.push_back(y);
        }
    for (Side& s : side) {
        const int cnt = static_cast<int>(s.pos.size());
        std::vector<double> cov(static_cast<size_t>(cnt) * cnt);
        for (int i = 0; i < cnt; ++i)
            for (int j = 0; j < cnt; ++j)
                cov[static_cast<size_t>(i) * cnt + j] =
                    covariance_model(channel, static_cast<double>(s.xs[i] - s.xs[j]), static_cast<double>(s.ys[i] - s.ys[j]));
        s.klt.resize(cov.size());
        klt_basis(cov.data(), cnt, s.klt.data());
    }
    const int keep0 = std::max(0, static_cast<int>(side[0].pos.size()) - 1);
    const int keep1 = std::max(0, static_cast<int>(side[1].pos.size()) - 1);
    int rows = 0;
    for (int i = 0; i < std::max(keep0, keep1); ++i) {
        if (i < keep0) {
            out.resize(out.size() + n);
            spread_row(side[0], i + 1, /*guard_zero_norm=*/false, n, out.data() + out.size() - n);
            ++rows;
        }
        if (i < keep1) {
            out.resize(out.size() + n);
            spread_row(side[1], i + 1, /*guard_zero_norm=*/true, n, out.data() + out.size() - n);
            ++rows;
        }
    }
    return rows;
}

}  // namespace

Dictionary build_dictionary(int bs) {
    Dictionary d;
    d.block_size = bs;
    d.n = bs * bs;
    d.cuts = distinct_line_cuts(bs);
    d.num_base = static_cast<int>(d.cuts.size());
    segment_atoms(bs, d.cuts, d.base);
    d.block_rows.assign(d.num_base, 0);
    d.block_row_off.assign(d.num_base + 1, 0);
    for (int ch = 0; ch < 3; ++ch) {
        d.detail[ch].clear();
        d.detail[ch].reserve(static_cast<size_t>(d.num_base) * d.n * d.n);
        for (int b = 0; b < d.num_base; ++b) {
            const int rows = detail_block(bs, d.cuts[b], ch, d.detail[ch]);
            d.block_rows[b] = rows;                  // identical for the three channels (shape-only)
        }
    }
    for (int b = 0; b < d.num_base; ++b) d.block_row_off[b + 1] = d.block_row_off[b] + d.block_rows[b];
    return d;
}

// ---------------------------------------------------------------------------
// Quantisation tables (reference: CompressedImage.cpp:16-166).  The three
// 32-entry variance tables and decay rates are the reference's calibration
// constants (Data/stats.txt 'variance' columns).
// ---------------------------------------------------------------------------
namespace {
const double kDecay[3] = {0.902045039488061, 0.896332644824969, 0.897340618787505};
const double kVariance[3][kMaxK] = {
    {1449455.61399403, 30867.8722232759, 4879.76236869648, 2065.81004100418, 1177.78544096912, 754.545827240537,
     519.229145237154, 375.509017928094, 281.02055698585, 216.291896608802, 170.433377219481, 137.390795170594,
     111.760859784514, 92.465892223227, 77.4762763021059, 65.5657474800836, 55.9694631412915, 48.1296222409122,
     41.6533463654379, 36.2103188098409, 31.620636785048, 27.6994619197188, 24.3350442530727, 21.416843198437,
     18.879120562683, 16.6557974119762, 14.7004164720848, 12.9791457307132, 11.4562911334688, 10.1096285629269,
     8.9223405657796, 7.86882234976579},
    {51995.6231219068, 814.727839313831, 108.677634702502, 49.7911952400269, 28.7429354781437, 19.0770865041308,
     13.7134473152652, 10.4220864760027, 8.24661056024952, 6.70462984374069, 5.55959248319828, 4.6929026981696,
     4.00644915447918, 3.4569689589025, 3.00626981236079, 2.63088816527246, 2.31483558804887, 2.04636560444097,
     1.81780225821739, 1.61760270635648, 1.44182998235134, 1.29183839884355, 1.15688638919911, 1.0404065817416,
     0.938042133945239, 0.844584472128933, 0.764573763603021, 0.691885170456949, 0.629607793503852,
     0.573021503073893, 0.52497651127712, 0.483288166342945},
    {60578.6241767756, 617.61939120778, 70.9553277465942, 31.4166652349442, 16.8206825627114, 10.4578171000126,
     7.14592512982323, 5.22663706003167, 4.01504652955091, 3.20518029980816, 2.64504709802794, 2.2477896896281,
     1.94630992425302, 1.72002308826788, 1.53680040297081, 1.39452712284538, 1.27393363160348, 1.17553215497528,
     1.08841824408173, 1.01505869329656, 0.950595903582374, 0.893571526043102, 0.841836849604898,
     0.792090537556817, 0.74756288452653, 0.704561007977196, 0.665399335594415, 0.631274499723472,
     0.597331898015673, 0.568852846586831, 0.539371538069597, 0.513182721162335}};
}  // namespace

void quantisation_tables(int K, int bs, double bpp, double* quant) {
    if (K < 1 || K > kMaxK) throw std::invalid_argument("K must be in 1..32");
    std::vector<double> bits(3 * static_cast<size_t>(K), 0.0), var(3 * static_cast<size_t>(K));
    for (int ch = 0; ch < 3; ++ch) std::copy(kVariance[ch], kVariance[ch] + K, var.begin() + ch * K);
    double allocated = 0.0;
    while ((allocated / static_cast<double>(bs * bs)) < bpp) {        // :138
        const ptrdiff_t idx = std::max_element(var.cbegin(), var.cend()) - var.cbegin();
        bits[idx] += 1.0;
        var[idx] /= 2.0;
        allocated += 1.0;
        const double floor_step = (idx % K == 0) ? static_cast<double>(bs) : 1.0;   // DC entries: index 0, K, 2K
        if (255.0 * static_cast<double>(bs) * std::pow(0.5, bits[idx]) < floor_step) var[idx] = 0.0;
    }
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) {
            const double floor_step = (i == 0) ? static_cast<double>(bs) : 1.0;
            const double step = std::ceil(255.0 * static_cast<double>(bs) * std::pow(kDecay[ch], static_cast<double>(i)) *
                                          std::pow(0.5, bits[ch * K + i]));
            quant[ch * K + i] = std::max(step, floor_step);
        }
}

std::vector<double> base_padded(const Dictionary& d, int pad_rows, int* padded_rows) {
    const int rows = ((d.num_base + pad_rows - 1) / pad_rows) * pad_rows;
    // one more zero row than reported: the kernel's scalar prefetch reads one row past the last swept one
    std::vector<double> out(static_cast<size_t>(rows + 1) * d.n, 0.0);
    std::copy(d.base.begin(), d.base.end(), out.begin());
    if (padded_rows) *padded_rows = rows;
    return out;
}

uint16_t bf16_round(float x) {
    uint32_t bits;
    std::memcpy(&bits, &x, sizeof(bits));
    bits += 0x7FFFu + ((bits >> 16) & 1u);
    return static_cast<uint16_t>(bits >> 16);
}

std::vector<uint16_t> filter_tiles(const double* rows, int nrows, int tiles, int k_order, std::vector<uint8_t>* shadow_out) {
    std::vector<char> shadowed(static_cast<size_t>(nrows), 0);
    for (int j = 1; j < nrows; ++j)
        for (int i = 0; i < j && !shadowed[j]; ++i) {
            bool same = true, negated = true;
            for (int k = 0; k < kTileN && (same || negated); ++k) {
                const double a = rows[static_cast<size_t>(i) * kTileN + k], b = rows[static_cast<size_t>(j) * kTileN + k];
                if (std::memcmp(&a, &b, sizeof(double)) != 0) same = false;
                const double nb = -b;
                if (std::memcmp(&a, &nb, sizeof(double)) != 0) negated = false;
            }
            if (same || negated) shadowed[j] = 1;
        }
    if (shadow_out) shadow_out->assign(shadowed.begin(), shadowed.end());
    std::vector<uint16_t> out(static_cast<size_t>(tiles) * kFilterTileHalves, 0);
    for (int tile = 0; tile < tiles; ++tile)
        for (int kk = 0; kk < 2; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int row = tile * 16 + (lane & 15);
                    const int hrow = lane >> 4;
                    const int k = k_order == 0 ? 32 * kk + 8 * hrow + j : 16 * (hrow ^ (hrow >> 1)) + 8 * kk + j;
                    if (row >= nrows || shadowed[row]) continue;
                    const float x = static_cast<float>(rows[static_cast<size_t>(row) * kTileN + k]);
                    const uint16_t hi = bf16_round(x);
                    const uint32_t hi_bits = static_cast<uint32_t>(hi) << 16;
                    float hi_f;
                    std::memcpy(&hi_f, &hi_bits, sizeof(hi_f));
                    const uint16_t lo = bf16_round(x - hi_f);
                    out[((static_cast<size_t>(tile) * 4 + 2 * kk + 0) * 64 + lane) * 8 + j] = hi;
                    out[((static_cast<size_t>(tile) * 4 + 2 * kk + 1) * 64 + lane) * 8 + j] = lo;
                }
    return out;
}

}  // namespace mpc
