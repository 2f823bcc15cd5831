// test_dropin.cpp -- a caller written like Compression/Compression.cpp (modes -c :96-128, -n :144-182, -s :200-264 of the
// reference), against the reference's interface as the drop-in (dropin/compressionlib_dropin.cpp) implements it.
//   test_dropin <mode> <width> <height> <seed> <bpp|max> <out file>
//     c   createCompressionContext -> encodeImage (double names) -> container written to <out file>
//     f   the same through the ...Fast names (what Compression.cpp itself calls)
//     n   encode + decodeImage + calculatePSNR, prints "PSNR <p> bytes <n>"; decoded RGB written to <out file>
//     s   64 random patches through CalcMPDynamic with the context's closures; "<count> <deltaId> <intCoeff> ..." per patch
//     x   a foreign closure (not made by the factory, same dictionary) is recognised by probing; a wrong one throws
//     h   host only (no GPU): the context's tables and closures through the reference's value types -- dictionary shapes, a
//         dynamic dictionary with repeats, FromCoeffsDynamic -- printed as text (two builds of this file, against the restated
//         and against the reference's real math:: / img:: headers, must print the same)
#include "CompressedImage.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <stdexcept>
#include <string>

using namespace compressed;
using namespace matching;
using namespace img;

constexpr size_t K = 32;
constexpr size_t BlockSize = 8;

// BASELINE.md section 3 generator: std::mt19937(seed), raster order, one draw per pixel
static std::unique_ptr<image<rgb>> synthetic(size_t W, size_t H, unsigned seed) {
    std::unique_ptr<image<rgb>> im = std::make_unique<image<rgb>>(W, H, false);
    std::mt19937 rng(seed);
    auto clamp = [](long v) { return static_cast<uchar>(v < 0 ? 0 : (v > 255 ? 255 : v)); };
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x) {
            const long n = static_cast<long>(rng() % 32) - 16;
            rgb px;
            px.r = clamp(static_cast<long>(x * 255 / W) + n);
            px.g = clamp(static_cast<long>(y * 255 / H) + n);
            px.b = clamp(128 + 3 * n);
            imRef(im, x, y) = px;
        }
    return im;
}

int main(int argc, char** argv) {
    if (argc != 7) return 2;
    const std::string mode = argv[1];
    const size_t W = std::strtoul(argv[2], nullptr, 10), H = std::strtoul(argv[3], nullptr, 10);
    const unsigned seed = static_cast<unsigned>(std::strtoul(argv[4], nullptr, 10));
    const std::string quality = argv[5];
    std::ofstream out(argv[6], std::ios::binary);
    try {
        std::unique_ptr<image<rgb>> imgIn = synthetic(W, H, seed);
        if (mode == "c" || mode == "n" || mode == "s" || mode == "x" || mode == "h") {
            std::unique_ptr<CompressionContext> context;
            if (quality == "max") {                                   // Compression.cpp:104-110
                context = createCompressionContext(K, BlockSize, 0.0);
                for (size_t i = 0; i < K; ++i) {
                    context->Y.Quant[i] = 1.0;
                    context->U.Quant[i] = 1.0;
                    context->V.Quant[i] = 1.0;
                }
            } else {
                context = createCompressionContext(K, BlockSize, std::atof(quality.c_str()));
            }
            if (mode == "h") {
                out.precision(17);
                out << context->K << ' ' << context->BlockSize << ' ' << context->BaseDict.Rows() << ' ' << context->BaseDict.Columns() << ' '
                    << context->Y.DetailBasis.size() << '\n';
                ChannelContext* chans[3] = {&context->Y, &context->U, &context->V};
                for (int ch = 0; ch < 3; ++ch) {
                    for (size_t i = 0; i < K; i += 7) out << chans[ch]->Quant[i] << ' ';
                    std::vector<BasisChoice> picks(4);
                    const unsigned short ids[4] = {3, static_cast<unsigned short>(2 * 40), static_cast<unsigned short>(2 * 37 + 1), 0};   // 3, +40, -38 (... 5), +0
                    for (int i = 0; i < 4; ++i) { picks[i].deltaId = ids[i]; picks[i].intCoeff = static_cast<unsigned short>(2 * (i + 1)); }
                    const math::Matrix dyn = chans[ch]->Dynamic(4, picks);
                    double sum = 0.0;
                    for (size_t r = 0; r < dyn.Rows(); ++r) sum += dyn.Data()[r * dyn.Columns() + (r % dyn.Columns())] * static_cast<double>(r % 5 + 1);
                    out << dyn.Rows() << ' ' << dyn.Columns() << ' ' << sum << ' ';
                    const math::Vector back = FromCoeffsDynamic(static_cast<int>(K), chans[ch]->Quant.Data(), picks, chans[ch]->Dynamic);
                    for (size_t j = 0; j < back.Length(); j += 9) out << back[j] << ' ';
                    out << '\n';
                }
                std::unique_ptr<image<rgb>> same = synthetic(W, H, seed);
                out << calculatePSNR(imgIn.get(), same.get()) << '\n';
                return 0;
            }
            if (mode == "s") {                                        // Compression.cpp:229-264
                std::mt19937 rng(seed);
                std::vector<BasisChoice> choices(K);
                ChannelContext* chans[3] = {&context->Y, &context->U, &context->V};
                for (int p = 0; p < 64; ++p) {
                    const size_t x = rng() % (imgIn->width() - BlockSize), y = rng() % (imgIn->height() - BlockSize);
                    for (int ch = 0; ch < 3; ++ch) {
                        math::Vector patch(BlockSize * BlockSize);
                        for (size_t offx = 0; offx < BlockSize; ++offx)
                            for (size_t offy = 0; offy < BlockSize; ++offy) {
                                const rgb pt = imRef(imgIn, x + offx, y + offy);
                                const double Y = (0.299 * pt.r + 0.587 * pt.g + 0.114 * pt.b);
                                const double v = ch == 0 ? Y : (ch == 1 ? ((0.436 / (1.0 - 0.114)) * (pt.b - Y)) : ((0.615 / (1.0 - 0.299)) * (pt.r - Y)));
                                patch[offx + offy * BlockSize] = v;
                            }
                        const int count = CalcMPDynamic(static_cast<int>(K), chans[ch]->Quant.Data(), choices, patch, chans[ch]->Dynamic);
                        out << count;
                        for (int i = 0; i < count; ++i) out << ' ' << choices[i].deltaId << ' ' << choices[i].intCoeff;
                        out << '\n';
                    }
                }
                return 0;
            }
            if (mode == "x") {
                // a closure the factory did not make, with the same behaviour: recognised by probing
                const DynamicDictionaryFunction inner[3] = {context->Y.Dynamic, context->U.Dynamic, context->V.Dynamic};
                DynamicDictionaryFunction foreign[3];
                for (int ch = 0; ch < 3; ++ch)
                    foreign[ch] = [&inner, ch](int reps, const std::vector<BasisChoice>& r) { return inner[ch](reps, r); };
                size_t n1 = 0, n2 = 0;
                std::unique_ptr<uint8_t[]> a = encodeImage(imgIn.get(), K, BlockSize, context->Y.Quant.Data(), context->U.Quant.Data(),
                                                           context->V.Quant.Data(), foreign[0], foreign[1], foreign[2], n1);
                std::unique_ptr<uint8_t[]> b = encodeImage(imgIn.get(), K, BlockSize, context->Y.Quant.Data(), context->U.Quant.Data(),
                                                           context->V.Quant.Data(), context->Y.Dynamic, context->U.Dynamic, context->V.Dynamic, n2);
                if (n1 != n2 || std::memcmp(a.get(), b.get(), n1) != 0) return 3;
                bool thrown = false;                                  // channels swapped: not the standard layout -> range_error*
                try {
                    encodeImage(imgIn.get(), K, BlockSize, context->Y.Quant.Data(), context->U.Quant.Data(), context->V.Quant.Data(),
                                foreign[1], foreign[0], foreign[2], n1);
                } catch (std::range_error* e) {
                    thrown = true;
                    delete e;
                }
                out << (thrown ? "ok" : "no throw");
                return thrown ? 0 : 4;
            }
            size_t outputBytesSize = 0;
            std::unique_ptr<uint8_t[]> encodedBytes = encodeImage(                       // Compression.cpp:112-123, double names
                imgIn.get(), K, BlockSize, context->Y.Quant.Data(), context->U.Quant.Data(), context->V.Quant.Data(),
                context->Y.Dynamic, context->U.Dynamic, context->V.Dynamic, outputBytesSize);
            if (mode == "c") {
                out.write(reinterpret_cast<const char*>(encodedBytes.get()), static_cast<std::streamsize>(outputBytesSize));
                return 0;
            }
            std::unique_ptr<image<rgb>> imgOut = decodeImage(encodedBytes.get(), outputBytesSize);      // :166-170
            const double psnr = calculatePSNR(imgIn.get(), imgOut.get());
            std::printf("PSNR %.10f bytes %zu\n", psnr, outputBytesSize);
            out.write(reinterpret_cast<const char*>(imgOut->data), static_cast<std::streamsize>(imgOut->width() * imgOut->height() * 3));
            return 0;
        }
        if (mode == "f") {                                            // Compression.cpp:102-123 verbatim names
            std::unique_ptr<CompressionContextFast> context;
            if (quality == "max") {
                context = createCompressionContextFast(K, BlockSize, 0.0);
                for (size_t i = 0; i < K; ++i) {
                    context->Y.Quant[i] = 1.0;
                    context->U.Quant[i] = 1.0;
                    context->V.Quant[i] = 1.0;
                }
            } else {
                context = createCompressionContextFast(K, BlockSize, std::atof(quality.c_str()));
            }
            size_t outputBytesSize = 0;
            std::unique_ptr<uint8_t[]> encodedBytes = encodeImageFast(imgIn.get(), K, BlockSize, context->Y.Quant, context->U.Quant,
                                                                      context->V.Quant, context->Y.Dynamic, context->U.Dynamic,
                                                                      context->V.Dynamic, outputBytesSize);
            out.write(reinterpret_cast<const char*>(encodedBytes.get()), static_cast<std::streamsize>(outputBytesSize));
            return 0;
        }
    } catch (std::range_error* e) {
        std::printf("range_error*: %s\n", e->what());
        delete e;
        return 5;
    }
    return 2;
}
