// test_mirror.cpp -- the reference's entropy-stage test properties (Testing/HuffmanTest.cpp: EmptyTest, BasicTest,
// BasicTest2, LargeTest, CorruptStreamTest, RLETest, LongSequences) written against the C++ mirror
// include/mpcodec.hpp, plus (with --gpu) an encodeImage / decodeImage / CalcMPDynamic round trip on the device.
// No gtest in the image: a tiny CHECK macro and a per-test runner.  Exit code 0 = all passed.
#include <mpcodec.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>

static int g_failed = 0;
#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);  \
            ++g_failed;                                                      \
            return;                                                          \
        }                                                                    \
    } while (0)
// the reference throws heap-allocated std::range_error (ASSERT_THROW(..., std::range_error*))
#define CHECK_THROWS_RANGE_ERROR_PTR(stmt)                 \
    do {                                                   \
        bool thrown = false;                               \
        try { stmt; } catch (std::range_error * e) {       \
            thrown = true;                                 \
            delete e;                                      \
        }                                                  \
        CHECK(thrown);                                     \
    } while (0)

static std::vector<uint16_t> roundtrip(const std::vector<uint16_t>& v) {
    size_t n = 0;
    std::unique_ptr<uint8_t[]> enc = huffman::huffmanEncode(v, n);
    return huffman::huffmanDecode(enc.get(), n);
}

static void HuffmanEmptyTest() {
    std::vector<uint16_t> empty;
    CHECK(roundtrip(empty) == empty);
}

static void HuffmanAllPartitionsOfTen() {
    // every integer partition of 10 as a symbol-frequency profile: many Huffman tree shapes
    const size_t n = 10;
    std::mt19937 rand;
    std::vector<uint16_t> counts(n + 1);
    size_t k = 1;
    counts[1] = n;
    int cases = 0;
    while (k != 0) {
        uint16_t x = counts[k - 1] + 1;
        uint16_t y = counts[k] - 1;
        --k;
        while (x <= y) { counts[k] = x; y -= x; ++k; }
        counts[k] = x + y;
        std::vector<uint16_t> test;
        for (size_t i = 0; i <= k; ++i)
            for (uint16_t c = 0; c < counts[i]; ++c) test.push_back(static_cast<uint16_t>(i));
        std::shuffle(test.begin(), test.end(), rand);
        CHECK(roundtrip(test) == test);
        ++cases;
    }
    CHECK(cases == 42);       // p(10)
}

static void HuffmanAllBitsSymbolVersusPseudoEof() {
    std::vector<uint16_t> test = {0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 15};
    CHECK(roundtrip(test) == test);
    const std::string text = "Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt ut labore";
    std::vector<uint16_t> test2(text.cbegin(), text.cend());
    CHECK(roundtrip(test2) == test2);
}

static void HuffmanLargeTest() {
    std::mt19937 rand;
    std::vector<uint16_t> test;
    for (size_t i = 0; i < 100000; ++i) {
        if ((rand() & 1) == 0) test.push_back(static_cast<uint16_t>(rand() % 0xFFFF));
        else test.push_back(static_cast<uint16_t>(100 + (rand() % 5)));
    }
    CHECK(roundtrip(test) == test);
}

static void HuffmanCorruptStreamTest() {
    const uint8_t nothing[1] = {0};
    CHECK_THROWS_RANGE_ERROR_PTR(huffman::huffmanDecode(nothing, 0));
    const std::string text = "Eleifend quam adipiscing vitae proin sagittis nisl rhoncus mattis rhoncus.";
    std::vector<uint16_t> test(text.cbegin(), text.cend());
    size_t n = 0;
    std::unique_ptr<uint8_t[]> enc = huffman::huffmanEncode(test, n);
    CHECK_THROWS_RANGE_ERROR_PTR(huffman::huffmanDecode(enc.get(), n / 3));      // truncated
}

static void RunLengthStrings() {
    const std::string values[] = {"", "a", "aa", "aaa", "aaaa", "aaaaa", "b", "ab", "aab", "aaab", "aaaab", "aaaaab",
                                  "bab", "baab", "baaab", "baaaab", "baaaaab", "aabbcc", "aaaabbbbcccc", "abababa",
                                  std::string(1, '\0'), std::string(2, '\0'), std::string(3, '\0'), std::string(4, '\0'),
                                  std::string(5, '\0')};
    for (const std::string& val : values) {
        std::vector<uint16_t> v(val.cbegin(), val.cend());
        CHECK(huffman::runLengthDecode(huffman::runLengthEncode(v)) == v);
    }
}

static void RunLengthLongSequences() {
    for (size_t len = 10; len < 50000; len += 997) {
        std::vector<uint16_t> v(len);
        CHECK(huffman::runLengthDecode(huffman::runLengthEncode(v)) == v);
        v[0] = 1;
        v[len - 1] = 1;
        CHECK(huffman::runLengthDecode(huffman::runLengthEncode(v)) == v);
    }
}

static void HostOnlyContextRefusesToEncode() {
    auto context = compressed::createCompressionContext(8, 8, 3.5, /*device*/ -1);
    CHECK(context->Y.Quant.size() == 8 && context->Y.Quant[0] == 8.0);     // DC step is capped at the block size
    img::image<img::rgb> im(16, 16);
    size_t n = 0;
    CHECK_THROWS_RANGE_ERROR_PTR(compressed::encodeImage(&im, *context, context->Y.Quant.data(), context->U.Quant.data(),
                                                         context->V.Quant.data(), n));
    CHECK_THROWS_RANGE_ERROR_PTR(compressed::createCompressionContext(64, 8, 3.5, -1));
}

// ---- device round trip (Compression.cpp -n mode: encode, decode, PSNR / bpp) ----
static void DeviceEncodeDecodeRoundTrip() {
    const size_t W = 200, H = 120, K = 32;
    auto context = compressed::createCompressionContext(K, 8, 3.5, 0);
    img::image<img::rgb> im(W, H);
    std::mt19937 rng(12345);
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x) {
            const int n = static_cast<int>(rng() % 32) - 16;
            auto clamp = [](int v) { return static_cast<img::uchar>(v < 0 ? 0 : (v > 255 ? 255 : v)); };
            imRef((&im), x, y) = img::rgb{clamp(static_cast<int>(x * 255 / W) + n), clamp(static_cast<int>(y * 255 / H) + n), clamp(128 + 3 * n)};
        }
    size_t bytes = 0;
    auto enc = compressed::encodeImage(&im, *context, context->Y.Quant.data(), context->U.Quant.data(), context->V.Quant.data(), bytes);
    CHECK(bytes > 14 + 192);
    auto dec = compressed::decodeImage(enc.get(), bytes, context.get());
    CHECK(dec->width() == W && dec->height() == H);
    const double psnr = compressed::calculatePSNR(&im, dec.get());
    const double bpp = 8.0 * static_cast<double>(bytes) / static_cast<double>(W * H);
    std::printf("  PSNR %.2f bpp %.3f bytes %zu\n", psnr, bpp, bytes);
    CHECK(psnr > 30.0 && bpp > 0.5 && bpp < 12.0);
    // encoding twice gives the same bytes
    size_t bytes2 = 0;
    auto enc2 = compressed::encodeImage(&im, *context, context->Y.Quant.data(), context->U.Quant.data(), context->V.Quant.data(), bytes2);
    CHECK(bytes2 == bytes && std::memcmp(enc.get(), enc2.get(), bytes) == 0);
    // CalcMPDynamic on one vector: a pure DC patch selects atom 0 first and stops when the residual quantises to 0
    std::vector<matching::BasisChoice> results(K);
    std::vector<double> patch(64, 100.0);
    const int count = matching::CalcMPDynamic(*context, 0, context->Y.Quant.data(), results, patch.data());
    CHECK(count >= 1 && results[0].deltaId == 0);
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::string(argv[1]) == "--gpu";
    struct { const char* name; void (*fn)(); bool needs_gpu; } tests[] = {
        {"HuffmanTests.EmptyTest", HuffmanEmptyTest, false},
        {"HuffmanTests.BasicTest", HuffmanAllPartitionsOfTen, false},
        {"HuffmanTests.BasicTest2", HuffmanAllBitsSymbolVersusPseudoEof, false},
        {"HuffmanTests.LargeTest", HuffmanLargeTest, false},
        {"HuffmanTests.CorruptStreamTest", HuffmanCorruptStreamTest, false},
        {"RunLengthTest.Strings", RunLengthStrings, false},
        {"RunLengthTest.LongSequences", RunLengthLongSequences, false},
        {"Context.HostOnlyRefusesToEncode", HostOnlyContextRefusesToEncode, false},
        {"Device.EncodeDecodeRoundTrip", DeviceEncodeDecodeRoundTrip, true},
    };
    int run = 0;
    for (auto& t : tests) {
        if (t.needs_gpu != gpu) continue;
        const int before = g_failed;
        std::printf("[ RUN  ] %s\n", t.name);
        try { t.fn(); } catch (std::range_error * e) {
            std::printf("  unexpected std::range_error*: %s\n", e->what());
            delete e;
            ++g_failed;
        }
        std::printf("[ %s ] %s\n", g_failed == before ? " OK " : "FAIL", t.name);
        ++run;
    }
    std::printf("%d tests, %d failed\n", run, g_failed);
    return g_failed ? 1 : 0;
}
