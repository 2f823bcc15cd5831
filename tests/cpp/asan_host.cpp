// asan_host.cpp -- the product's HOST code under AddressSanitizer + UBSan (g++, no GPU, no HIP): `make asan` / tests/test_asan_host.py.
// The host sources are compiled into this one translation unit and driven through what parses untrusted data or sizes buffers:
// dictionary and quantiser tables, Huffman / RLE / container round trips on random records, and a corpus of truncated and
// bit-flipped containers (small ones made here, plus prefixes of the reference's 3.7 MB .mn when its path is given).
#include "../../imageexperiments_amd/csrc/host_bitstream.cpp"
#include "../../imageexperiments_amd/csrc/host_codec.cpp"
#include "../../imageexperiments_amd/csrc/host_dictionary.cpp"
#include "../../imageexperiments_amd/csrc/host_stats.cpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <random>

static int g_failed = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            ++g_failed;                                                     \
        }                                                                   \
    } while (0)

static std::vector<uint8_t> random_container(std::mt19937& rng, int W, int H, int K) {
    const int tx = (W + 7) / 8, ty = (H + 7) / 8;
    const size_t tiles = static_cast<size_t>(tx) * ty;
    std::vector<uint16_t> counts(3 * tiles);
    std::vector<uint32_t> choices(3 * tiles * K, 0);
    for (size_t o = 0; o < 3 * tiles; ++o) {
        counts[o] = static_cast<uint16_t>(rng() % (K + 1));
        for (int i = 0; i < counts[o]; ++i)
            choices[o * K + i] = (rng() % 2500) | ((rng() % (i == 0 ? 4000 : 60)) << 16);
    }
    std::vector<double> quant(3 * static_cast<size_t>(K));
    mpc::quantisation_tables(K, 8, 3.5, quant.data());
    std::vector<uint8_t> blob = mpc::encode_records(W, H, K, 8, quant.data(), counts.data(), choices.data());
    mpc::Streams s;
    CHECK(mpc::read_compressed(blob.data(), blob.size(), s));
    std::vector<uint16_t> c2;
    std::vector<uint32_t> h2;
    CHECK(mpc::disassemble_streams(s, c2, h2));
    CHECK(c2 == counts);
    bool same = true;
    for (size_t o = 0; o < 3 * tiles && same; ++o)
        for (int i = 0; i < counts[o] && same; ++i) same = h2[o * K + i] == choices[o * K + i];
    CHECK(same);
    // the device-assembled form of the same streams gives the same bytes
    std::vector<uint16_t> symbols;
    std::vector<unsigned long long> off(6 * static_cast<size_t>(K) + 1, 0);
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i)
            for (int part = 0; part < 2; ++part) {
                off[2 * (ch * K + i) + part] = symbols.size();
                int32_t prev = 0;
                for (size_t t = 0; t < tiles; ++t)
                    if (counts[3 * t + ch] > i) {
                        const uint32_t rec = choices[(3 * t + ch) * K + i];
                        uint16_t v = part == 0 ? static_cast<uint16_t>(rec & 0xFFFF) : static_cast<uint16_t>(rec >> 16);
                        if (part == 1 && i == 0) {
                            const int32_t d = static_cast<int32_t>(v) - prev;
                            prev = v;
                            v = static_cast<uint16_t>(mpc::zigzag_encode(d));
                        }
                        symbols.push_back(v);
                    }
            }
    off[6 * static_cast<size_t>(K)] = symbols.size();
    size_t n2 = 0;
    uint8_t* b2 = mpc::encode_symbol_streams_malloc(W, H, K, 8, quant.data(), counts.data(), symbols.data(), off.data(), &n2);
    CHECK(b2 && n2 == blob.size() && std::memcmp(b2, blob.data(), n2) == 0);
    std::free(b2);
    return blob;
}

static void fuzz(const std::vector<uint8_t>& blob, std::mt19937& rng, int truncations, int flips) {
    for (int k = 0; k < truncations; ++k) {
        const size_t cut = blob.empty() ? 0 : rng() % blob.size();
        std::vector<uint8_t> part(blob.begin(), blob.begin() + cut);   // exact-size heap buffer: a read past the end is an ASan report
        mpc::Streams s;
        (void)mpc::read_compressed(part.data(), part.size(), s);
    }
    for (int k = 0; k < flips; ++k) {
        std::vector<uint8_t> bad(blob);
        for (int f = 0; f < 1 + static_cast<int>(rng() % 3); ++f) bad[rng() % bad.size()] ^= static_cast<uint8_t>(1u << (rng() % 8));
        mpc::Streams s;
        if (mpc::read_compressed(bad.data(), bad.size(), s)) {
            std::vector<uint16_t> c;
            std::vector<uint32_t> h;
            (void)mpc::disassemble_streams(s, c, h);
        }
    }
}

int main(int argc, char** argv) {
    std::mt19937 rng(20260401);
    {   // dictionary and tables (the eigen solver, the line cuts, the bit allocation)
        const mpc::Dictionary d8 = mpc::build_dictionary(8);
        CHECK(d8.num_base == 510 && d8.total_detail_rows() == 31622);
        const mpc::Dictionary d4 = mpc::build_dictionary(4);
        CHECK(d4.num_base > 0);
        for (int K : {1, 8, 32})
            for (double bpp : {0.0, 2.0, 3.5, 6.0}) {
                std::vector<double> q(3 * static_cast<size_t>(K));
                mpc::quantisation_tables(K, 8, bpp, q.data());
                CHECK(q[0] >= 1.0);
            }
        std::vector<uint8_t> shadow;
        const std::vector<uint16_t> t = mpc::filter_tiles(d8.base.data(), d8.num_base, 32, 1, &shadow);
        CHECK(t.size() == 32u * 2048u && shadow.size() == 510u && shadow[509] == 1);
    }
    {   // the worker pool contains a throwing job: the call drains, the first exception reaches the caller, the pool stays usable
        for (int rep = 0; rep < 20; ++rep) {
            std::atomic<int> ran{0};
            bool caught = false;
            try {
                mpc::parallel_jobs(200, [&](int i) {
                    ++ran;
                    if (i == 37 + rep) throw std::runtime_error("job failed");
                });
            } catch (const std::runtime_error& e) {
                caught = std::strcmp(e.what(), "job failed") == 0;
            }
            CHECK(caught);
            CHECK(ran.load() >= 1 && ran.load() <= 200);
            std::atomic<int> sum{0};
            mpc::parallel_jobs(200, [&](int i) { sum += i; });
            CHECK(sum.load() == 199 * 200 / 2);
        }
    }
    {   // Huffman and run-length codes
        for (int rep = 0; rep < 60; ++rep) {
            std::vector<uint16_t> v(rng() % 5000);
            const unsigned range = 1u << (rng() % 16);
            for (uint16_t& x : v) x = static_cast<uint16_t>((rng() % 4 == 0) ? rng() % 65536 : rng() % range);
            mpc::BitWriter w;
            mpc::huffman_encode(v.data(), v.size(), w);
            const std::vector<uint8_t> bytes = w.bytes();
            mpc::BitReader r(bytes.data(), bytes.size());
            std::vector<uint16_t> back;
            CHECK(mpc::huffman_decode(r, back) && back == v);
            const std::vector<uint16_t> packed = mpc::rle_encode(v.data(), v.size());
            CHECK(mpc::rle_decode(packed.data(), packed.size()) == v);
            for (int k = 0; k < 20 && !bytes.empty(); ++k) {           // corrupt Huffman streams
                std::vector<uint8_t> bad(bytes.begin(), bytes.begin() + (rng() % bytes.size()));
                if (!bad.empty()) bad[rng() % bad.size()] ^= 0x10;
                mpc::BitReader rb(bad.data(), bad.size());
                std::vector<uint16_t> junk;
                (void)mpc::huffman_decode(rb, junk);
            }
        }
    }
    for (auto shape : {std::pair<int, int>{40, 24}, {129, 77}, {8, 8}, {3, 5}})
        for (int K : {1, 5, 32}) {
            const std::vector<uint8_t> blob = random_container(rng, shape.first, shape.second, K);
            fuzz(blob, rng, 60, 120);
        }
    {   // the planned route of the entropy stage (plan_stream / or_bits: the host's half of the device-side stage) against the direct one
        for (int rep = 0; rep < 12; ++rep) {
            const int K = 1 + static_cast<int>(rng() % 4), W = 64, H = 40;
            const size_t tiles = static_cast<size_t>(W / 8) * (H / 8);
            std::vector<uint16_t> counts(3 * tiles);
            for (uint16_t& c : counts) c = static_cast<uint16_t>(rng() % (K + 1));
            std::vector<unsigned long long> off(6 * static_cast<size_t>(K) + 1, 0);
            std::vector<uint16_t> symbols;
            for (int st = 0; st < 6 * K; ++st) {
                const size_t n = rng() % 3 == 0 ? 0 : rng() % 70000;
                const unsigned range = 1u << (rng() % 16);
                uint16_t v = 0;
                for (size_t i = 0; i < n; ++i) {
                    if (rng() % 8 != 0 || i == 0) v = static_cast<uint16_t>(rng() % range);       // runs of equal symbols
                    if (rng() % 5000 == 0) { symbols.insert(symbols.end(), 40000, v); i += 40000; }   // long enough to be cut at 0x8001
                    symbols.push_back(v);
                }
                off[static_cast<size_t>(st) + 1] = symbols.size();
            }
            std::vector<double> q(3 * static_cast<size_t>(K), 8.0);
            size_t n1 = 0, n2 = 0;
            uint8_t* direct = mpc::encode_symbol_streams_malloc(W, H, K, 8, q.data(), counts.data(), symbols.data(), off.data(), &n1);
            uint8_t* planned = mpc::encode_symbol_streams_by_plan_malloc(W, H, K, 8, q.data(), counts.data(), symbols.data(), off.data(), &n2);
            CHECK(direct && planned && n1 == n2 && std::memcmp(direct, planned, n1) == 0);
            std::free(direct);
            std::free(planned);
        }
    }
    if (argc > 1) {                                                   // the reference's own bitstream
        std::ifstream f(argv[1], std::ios::binary);
        std::vector<uint8_t> mn((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        CHECK(!mn.empty());
        mpc::Streams s;
        CHECK(mpc::read_compressed(mn.data(), mn.size(), s));
        CHECK(mpc::write_compressed(s) == mn);
        std::vector<uint8_t> head(mn.begin(), mn.begin() + std::min<size_t>(mn.size(), 200000));
        fuzz(head, rng, 12, 12);
    }
    std::printf("asan_host: %d failed\n", g_failed);
    return g_failed ? 1 : 0;
}
