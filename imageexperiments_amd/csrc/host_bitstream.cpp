// host_bitstream.cpp -- product host code: entropy stage + container (see host_bitstream.h).
#include "host_bitstream.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <numeric>
#include <thread>

namespace mpc {

// ------------------------------------------------------------------------------------------------
// bits
// ------------------------------------------------------------------------------------------------
void BitWriter::put(uint64_t value, int width) {
    if (width <= 0) return;
    if (width < 64) value &= (1ULL << width) - 1ULL;
    const size_t word = nbits_ >> 6;
    const int used = static_cast<int>(nbits_ & 63);
    if (words_.size() < word + 2) words_.resize(word + 2, 0);
    const int room = 64 - used;
    if (width <= room) {
        words_[word] |= (width == 64) ? value : (value << (room - width));
    } else {
        words_[word] |= value >> (width - room);
        words_[word + 1] = value << (64 - (width - room));
    }
    nbits_ += static_cast<size_t>(width);
}

void BitWriter::put_codes(const uint16_t* data, size_t n, const uint32_t* code_of, const uint8_t* length_of, size_t total_bits) {
    if (n == 0) return;
    words_.resize(((nbits_ + total_bits) >> 6) + 2, 0);       // everything this call writes, plus put()'s spare word
    size_t word = nbits_ >> 6;
    int used = static_cast<int>(nbits_ & 63);
    uint64_t acc = used ? words_[word] >> (64 - used) : 0;     // the bits already in the current word, right-aligned
    uint64_t* const dst = words_.data();
    auto emit = [&](uint64_t code, int len) {                  // len 1..64
        if (used + len <= 64) {
            acc = (len == 64) ? code : ((acc << len) | code);
            used += len;
            if (used == 64) { dst[word++] = acc; acc = 0; used = 0; }
        } else {
            const int spill = used + len - 64;                 // low `spill` bits go to the next word
            dst[word++] = (acc << (len - spill)) | (code >> spill);
            acc = code & ((1ULL << spill) - 1ULL);
            used = spill;
        }
    };
    size_t i = 0;
    for (; i + 1 < n; i += 2) {                                 // two codes (<= 32 bits each) joined into one emission
        const uint16_t a = data[i], b = data[i + 1];
        const int lb = length_of[b];
        emit((static_cast<uint64_t>(code_of[a]) << lb) | code_of[b], length_of[a] + lb);
    }
    if (i < n) emit(code_of[data[i]], length_of[data[i]]);
    dst[word] = used ? acc << (64 - used) : 0;
    nbits_ += total_bits;
}

void BitWriter::append(const BitWriter& other) {
    if (other.nbits_ == 0) return;
    const size_t src_words = (other.nbits_ + 63) >> 6;
    const int used = static_cast<int>(nbits_ & 63);
    size_t word = nbits_ >> 6;
    words_.resize(((nbits_ + other.nbits_) >> 6) + 2, 0);
    if (used == 0) {
        std::memcpy(words_.data() + word, other.words_.data(), src_words * sizeof(uint64_t));
    } else {                                                   // every source word straddles two destination words
        uint64_t carry = words_[word];
        for (size_t i = 0; i < src_words; ++i) {
            const uint64_t v = other.words_[i];
            words_[word++] = carry | (v >> used);
            carry = v << (64 - used);
        }
        words_[word] = carry;
    }
    nbits_ += other.nbits_;
    // bits past the end must stay zero (put() ORs into the current word): the source's last word is zero-padded already,
    // but clear what the shift may have left in the word after the new end
    const size_t last = nbits_ >> 6;
    const int tail = static_cast<int>(nbits_ & 63);
    words_[last] = tail ? (words_[last] & ~((1ULL << (64 - tail)) - 1ULL)) : 0;
    for (size_t i = last + 1; i < words_.size(); ++i) words_[i] = 0;
}

std::vector<uint8_t> BitWriter::bytes() const {
    const size_t n = (nbits_ + 7) / 8;
    std::vector<uint8_t> out(n);
    const size_t whole = n / 8;
    for (size_t i = 0; i < whole; ++i) {
        const uint64_t be = __builtin_bswap64(words_[i]);       // MSB-first bit order = big-endian bytes
        std::memcpy(out.data() + 8 * i, &be, 8);
    }
    for (size_t i = 8 * whole; i < n; ++i) out[i] = static_cast<uint8_t>(words_[i >> 3] >> (56 - 8 * (i & 7)));
    return out;
}

uint64_t BitReader::get(int width) {
    if (width <= 0) return 0;
    const size_t left = nbits_ - pos_;
    if (static_cast<size_t>(width) > left) width = static_cast<int>(left);
    if (width <= 56 && (pos_ >> 3) + 8 <= ((nbits_ + 7) >> 3)) {      // one unaligned 64-bit load covers offset + width
        uint64_t w;
        std::memcpy(&w, p_ + (pos_ >> 3), 8);
        w = __builtin_bswap64(w) << (pos_ & 7);
        pos_ += static_cast<size_t>(width);
        return width ? w >> (64 - width) : 0;
    }
    uint64_t v = 0;
    int need = width;
    while (need > 0) {
        const size_t byte = pos_ >> 3;
        const int off = static_cast<int>(pos_ & 7);
        const int take = std::min(need, 8 - off);
        const uint32_t bits = (static_cast<uint32_t>(p_[byte]) >> (8 - off - take)) & ((1u << take) - 1u);
        v = (v << take) | bits;
        pos_ += static_cast<size_t>(take);
        need -= take;
    }
    return v;
}

uint32_t BitReader::peek32() const {
    const size_t byte = pos_ >> 3;
    const int off = static_cast<int>(pos_ & 7);
    const size_t nbytes = (nbits_ + 7) >> 3;
    if (byte + 8 <= nbytes) {
        uint64_t w;
        std::memcpy(&w, p_ + byte, 8);
        return static_cast<uint32_t>((__builtin_bswap64(w) << off) >> 32);
    }
    uint64_t window = 0;                                        // 5 bytes cover 32 bits at any bit offset
    if (byte + 5 <= nbytes) {
        window = (static_cast<uint64_t>(p_[byte]) << 32) | (static_cast<uint64_t>(p_[byte + 1]) << 24) |
                 (static_cast<uint64_t>(p_[byte + 2]) << 16) | (static_cast<uint64_t>(p_[byte + 3]) << 8) | p_[byte + 4];
    } else {
        for (int k = 0; k < 5; ++k) window = (window << 8) | (byte + k < nbytes ? p_[byte + k] : 0u);
    }
    return static_cast<uint32_t>(window >> (8 - off));
}

namespace {

inline uint32_t bit_width(uint32_t v) { return v ? 32u - static_cast<uint32_t>(__builtin_clz(v)) : 0u; }

}  // namespace

// BitBuffer.cpp:228-269 (the reference's own Golomb variant: remainder in bit_width(M) bits)
uint32_t golomb_length(uint32_t value, uint32_t m) {
    const uint32_t q = static_cast<uint32_t>(static_cast<int32_t>(value) / static_cast<int32_t>(m));
    const uint32_t rem = static_cast<uint32_t>(static_cast<int32_t>(value) % static_cast<int32_t>(m));
    const uint32_t b = bit_width(m);
    const uint32_t limit = (1u << (b + 1)) - m;
    return (rem < limit) ? b + q + 1 : b + q + 2;
}

void golomb_write(uint32_t value, uint32_t m, BitWriter& out) {
    uint32_t q = static_cast<uint32_t>(static_cast<int32_t>(value) / static_cast<int32_t>(m));
    const uint32_t rem = static_cast<uint32_t>(static_cast<int32_t>(value) % static_cast<int32_t>(m));
    while (q >= 64) { out.put(~0ULL, 64); q -= 64; }
    if (q) out.put((1ULL << q) - 1ULL, static_cast<int>(q));
    out.put(0, 1);
    const uint32_t b = bit_width(m);
    const uint32_t limit = (1u << (b + 1)) - m;
    if (rem < limit) out.put(rem, static_cast<int>(b));
    else out.put(static_cast<uint64_t>(rem) + limit, static_cast<int>(b) + 1);
}

uint32_t golomb_read(uint32_t m, BitReader& in) {
    uint32_t q = 0;
    // the unary part, 32 bits at a time while they are all ones (same result as reading bit by bit: past the end reads as 0)
    for (;;) {
        const size_t left = in.remaining();
        if (left == 0) break;
        const uint32_t ones = static_cast<uint32_t>(__builtin_clz(~in.peek32() | 1u));      // leading ones, 31 at most counted here
        if (ones >= 31 && left > 31) { q += 31; in.skip(31); continue; }
        const uint32_t take = static_cast<uint32_t>(std::min<size_t>(ones, left));
        q += take;
        in.skip(take);
        if (take < left) in.skip(1);                            // the terminating zero
        break;
    }
    const uint32_t b = bit_width(m);
    const uint32_t limit = (1u << (b + 1)) - m;
    const uint32_t first = static_cast<uint32_t>(in.get(static_cast<int>(b)));
    const uint32_t rem = (first < limit) ? first : (first << 1) + static_cast<uint32_t>(in.get(1)) - limit;
    return q * m + rem;
}

namespace {

// Elias-Fano coding of a non-decreasing u16 sequence (BitBuffer.cpp:292-354)
}  // namespace

uint32_t elias_fano_length(size_t n, uint16_t max_symbol) {
    if (n == 0) return 0;
    const uint32_t m = bit_width(max_symbol), nb = bit_width(static_cast<uint32_t>(n));
    const uint32_t low = (m >= nb) ? m - nb : 0;
    return static_cast<uint32_t>(n) * (1 + low) + (1u << (m - low)) - 1;
}

void elias_fano_write(const uint16_t* seq, size_t n, uint16_t max_symbol, BitWriter& out) {
    if (n == 0) return;
    const uint32_t m = bit_width(max_symbol), nb = bit_width(static_cast<uint32_t>(n));
    const uint32_t low = (m >= nb) ? m - nb : 0;
    uint32_t bucket = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t target = static_cast<uint16_t>(seq[i] >> low);
        while (bucket != target) { out.put(0, 1); ++bucket; }
        out.put(1, 1);
    }
    for (size_t i = 0; i < n; ++i) out.put(seq[i], static_cast<int>(low));
}

bool elias_fano_read(uint16_t* dst, size_t n, uint16_t max_symbol, BitReader& in) {
    if (n == 0) return true;
    const uint32_t m = bit_width(max_symbol), nb = bit_width(static_cast<uint32_t>(n));
    const uint32_t low = (m >= nb) ? m - nb : 0;
    uint16_t bucket = 0;
    for (size_t i = 0; i < n; ++i) {
        if (in.remaining() == 0) return false;
        while (in.get(1) == 0) {
            ++bucket;
            if (in.remaining() == 0) return false;
        }
        dst[i] = static_cast<uint16_t>(bucket << low);
    }
    for (size_t i = 0; i < n; ++i) dst[i] |= static_cast<uint16_t>(in.get(static_cast<int>(low)));
    return true;
}

namespace {
// ------------------------------------------------------------------------------------------------
// Iteration order of MSVC's std::unordered_map<uint32_t, T> after a given insertion sequence.
//
// huffmanEncode pushes its leaves into the priority queue while iterating such a map (Huffman.cpp:47,
// 72-81), so equal-frequency ties -- and with them code lengths and bytes -- follow that order.  The
// reference builds only with Visual Studio 2022; its STL keeps all elements in one doubly linked list plus
// [first,last] iterators per bucket (bucket = FNV-1a(key) & (buckets-1); 8 buckets, load factor 1, growth
// x8 below 512 buckets and to the next power of two >= size afterwards).  A key whose bucket is empty is
// linked at the end of the list, otherwise directly in front of its bucket's first element; a rehash walks
// the list in order applying the same rule.  Iteration = list order.
// ------------------------------------------------------------------------------------------------
class MsvcHashOrder {
public:
    MsvcHashOrder() : first_(8, kNone), last_(8, kNone) {
        next_.push_back(0);
        prev_.push_back(0);
        keys_.push_back(0);
    }
    // insert a key known to be absent; returns its node id (1-based, in insertion order)
    int insert(uint32_t key) {
        const int node = static_cast<int>(keys_.size());
        keys_.push_back(key);
        next_.push_back(0);
        prev_.push_back(0);
        if (size_ + 1 > first_.size()) grow(size_ + 1);
        place(node);
        ++size_;
        return node;
    }
    template <class F>
    void for_each(F&& f) const {
        for (int n = next_[0]; n != 0; n = next_[n]) f(n);
    }

private:
    static constexpr int kNone = 0;      // node 0 is the list head, i.e. end()
    static uint64_t fnv1a(uint32_t key) {
        uint64_t h = 14695981039346656037ULL;
        for (int i = 0; i < 4; ++i) {
            h ^= (key >> (8 * i)) & 0xFFu;
            h *= 1099511628211ULL;
        }
        return h;
    }
    size_t bucket_of(uint32_t key) const { return static_cast<size_t>(fnv1a(key)) & (first_.size() - 1); }
    void link_before(int node, int where) {
        const int before = prev_[where];
        next_[node] = where;
        prev_[node] = before;
        next_[before] = node;
        prev_[where] = node;
    }
    void unlink(int node) {
        next_[prev_[node]] = next_[node];
        prev_[next_[node]] = prev_[node];
    }
    void place(int node) {                      // node not linked yet
        const size_t b = bucket_of(keys_[node]);
        if (first_[b] == kNone) {
            link_before(node, 0);
            first_[b] = last_[b] = node;
        } else {
            link_before(node, first_[b]);
            first_[b] = node;
        }
    }
    void grow(size_t for_size) {
        size_t want = first_.size() < 512 && first_.size() * 8 >= for_size ? first_.size() * 8 : for_size;
        size_t buckets = 1;
        while (buckets < want) buckets <<= 1;
        first_.assign(buckets, kNone);
        last_.assign(buckets, kNone);
        int node = next_[0];
        while (node != 0) {
            const int following = next_[node];
            const size_t b = bucket_of(keys_[node]);
            if (first_[b] == kNone) {
                first_[b] = last_[b] = node;            // stays where it is
            } else {
                unlink(node);
                link_before(node, first_[b]);
                first_[b] = node;
            }
            node = following;
        }
    }
    std::vector<uint32_t> keys_;
    std::vector<int> next_, prev_, first_, last_;
    size_t size_ = 0;
};

constexpr uint32_t kPseudoEof = 0xFFFFFFFFu;

// The reference's std::priority_queue<Node*, vector, NodeCompare> (Huffman.cpp:21-25, 72-97) orders by frequency alone, so which
// of several equally frequent nodes comes out first is decided by the heap algorithm itself -- and with it the code lengths.
// This is std::push_heap / std::pop_heap (libstdc++'s sift-to-the-bottom-then-up, which reproduces the reference's bytes: the
// re-encode of Data/r0c1de5e1t_3_5.mn), step for step, on one packed word per node instead of a 16-byte struct behind
// iterator templates: building the 193 trees of a 16 Mpixel frame was half of the host's 8.5 ms of table building.
class FrequencyHeap {
public:
    static constexpr int kIdBits = 24;                           // ids < 2 * 65537; frequencies < 2^40
    void reserve(size_t n) { v_.reserve(n); }
    size_t size() const { return v_.size(); }
    void push(uint64_t freq, int id) {
        const uint64_t x = (freq << kIdBits) | static_cast<uint64_t>(id);
        v_.push_back(x);
        sift_up(v_.size() - 1, x);
    }
    // pop_heap + back() + pop_back(): the least frequent node
    void pop(uint64_t* freq, int* id) {
        const uint64_t top = v_.front(), value = v_.back();
        v_.pop_back();
        const size_t len = v_.size();
        if (len != 0) {
            uint64_t* v = v_.data();
            size_t hole = 0, child = 0;
            while (child < (len - 1) / 2) {
                child = 2 * (child + 1);
                child -= static_cast<size_t>(after(v[child], v[child - 1]));      // no branch: which child wins is a coin toss
                v[hole] = v[child];
                hole = child;
            }
            if ((len & 1) == 0 && child == (len - 2) / 2) {
                child = 2 * (child + 1);
                v[hole] = v[child - 1];
                hole = child - 1;
            }
            sift_up(hole, value);
        }
        *freq = top >> kIdBits;
        *id = static_cast<int>(top & ((1u << kIdBits) - 1u));
    }
private:
    static bool after(uint64_t a, uint64_t b) { return (a >> kIdBits) > (b >> kIdBits); }      // NodeCompare
    void sift_up(size_t hole, uint64_t x) {
        uint64_t* v = v_.data();
        while (hole > 0) {
            const size_t parent = (hole - 1) / 2;
            if (!after(v[parent], x)) break;
            v[hole] = v[parent];
            hole = parent;
        }
        v[hole] = x;
    }
    std::vector<uint64_t> v_;
};

struct Entry {
    uint32_t symbol;
    uint8_t length;
    uint32_t code;
};

}  // namespace

namespace {
// One pass over a stream: histogram (only as large as the largest symbol: no 64K-entry tables to clear per stream) and
// the distinct symbols in order of first appearance -- the order the reference inserts them into its unordered_map.
struct SymbolStats {
    uint16_t largest = 0;
    std::vector<uint32_t> hist;
    std::vector<uint16_t> distinct;
};

SymbolStats gather_stats(const uint16_t* data, size_t n) {
    SymbolStats st;
    uint16_t largest = 0;
    for (size_t i = 0; i < n; ++i) largest = std::max(largest, data[i]);       // vectorises
    st.largest = largest;
    const size_t bins = static_cast<size_t>(largest) + 1;
    st.hist.assign(bins, 0);
    if (n >= 4096 && bins <= 4096) {
        // long stream, few symbols: consecutive increments of one counter wait for each other (store forwarding); four
        // interleaved tables break the chain
        thread_local std::vector<uint32_t> lanes;
        lanes.assign(4 * bins, 0);
        uint32_t* h0 = lanes.data();
        uint32_t* h1 = h0 + bins;
        uint32_t* h2 = h1 + bins;
        uint32_t* h3 = h2 + bins;
        size_t i = 0;
        for (; i + 3 < n; i += 4) { ++h0[data[i]]; ++h1[data[i + 1]]; ++h2[data[i + 2]]; ++h3[data[i + 3]]; }
        for (; i < n; ++i) ++h0[data[i]];
        size_t present = 0;
        for (size_t b = 0; b < bins; ++b) {
            st.hist[b] = h0[b] + h1[b] + h2[b] + h3[b];
            present += st.hist[b] != 0;
        }
        // order of first appearance: scan until every symbol that occurs has been seen
        std::vector<uint8_t> seen(bins, 0);
        st.distinct.reserve(present);
        for (size_t k = 0; k < n && st.distinct.size() < present; ++k)
            if (!seen[data[k]]) { seen[data[k]] = 1; st.distinct.push_back(data[k]); }
    } else {
        for (size_t i = 0; i < n; ++i)
            if (st.hist[data[i]]++ == 0) st.distinct.push_back(data[i]);
    }
    return st;
}

// Number of symbols runLengthEncode would emit, without emitting them: a maximal run of L equal symbols becomes 1 symbol
// (L = 1) or 3 (symbol, symbol, L - 2); runs beyond 0x8001 symbols are cut by the reference's counter limit (Huffman.cpp:
// 262-267) -- if any run can be that long the literal state machine decides.
size_t rle_encoded_size(const uint16_t* data, size_t n) {
    if (n == 0) return 0;
    size_t equal = 0, long_starts = 0;                          // data[i] == data[i-1]; ... and data[i-1] != data[i-2]
    for (size_t i = 1; i < n; ++i) {
        const bool e = data[i] == data[i - 1];
        const bool before = i >= 2 && data[i - 1] == data[i - 2];
        equal += e;
        long_starts += e && !before;
    }
    if (equal < 0x8000) return (n - equal) + 2 * long_starts;
    size_t packed = 0;
    uint16_t prev = 0, run = 0;
    bool fresh = true;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t v = data[i];
        if (v == prev && !fresh) {
            if (++run == 1) ++packed;
            else if (run >= 0x8000) { ++packed; run = 0; fresh = true; }
        } else {
            fresh = false;
            if (run > 0) { ++packed; run = 0; }
            prev = v;
            ++packed;
        }
    }
    if (run > 0) ++packed;
    return packed;
}

void huffman_encode_with(const SymbolStats& st, const uint16_t* data, size_t n, BitWriter& out);
}  // namespace

// Huffman.cpp:46-163
void huffman_encode(const uint16_t* data, size_t n, BitWriter& out) { huffman_encode_with(gather_stats(data, n), data, n, out); }

namespace {
// Everything of huffmanEncode that does not touch the symbols one by one: tree, canonical codes, the table header
// written to `out`; codes and lengths by symbol (dense up to st.largest), the pseudo-EOF code, and the payload's bits.
struct HuffmanTable {
    std::vector<uint32_t> code_of;
    std::vector<uint8_t> length_of;
    uint32_t eof_code = 0;
    uint8_t eof_length = 0;
    uint8_t max_length = 0;
    size_t payload_bits = 0;
};

void huffman_table(const SymbolStats& st, size_t n, BitWriter& out, HuffmanTable& t) {
    const size_t table_size = static_cast<size_t>(st.largest) + 1;
    std::vector<uint32_t> symbols(st.distinct.begin(), st.distinct.end());
    std::vector<size_t> freq;
    freq.reserve(symbols.size() + 1);
    MsvcHashOrder order;
    for (uint32_t s : symbols) {
        freq.push_back(st.hist[s]);
        order.insert(s);                                       // node id == leaf + 1
    }
    const uint8_t symbol_bits = std::max<uint8_t>(1, static_cast<uint8_t>(bit_width(st.largest)));
    const int eof_leaf = static_cast<int>(symbols.size());
    symbols.push_back(kPseudoEof);
    freq.push_back(0);
    order.insert(kPseudoEof);
    const int leaves = static_cast<int>(symbols.size());

    // tree: leaves 0..leaves-1, internal nodes appended; code length = depth below the root
    std::vector<int> parent(static_cast<size_t>(2 * leaves), -1);
    std::vector<uint8_t> height(static_cast<size_t>(2 * leaves), 0);     // of the subtree below a node
    FrequencyHeap heap;
    heap.reserve(static_cast<size_t>(leaves));
    order.for_each([&](int node) { heap.push(freq[node - 1], node - 1); });
    int next_id = leaves;
    while (heap.size() > 1) {
        uint64_t fa, fb;
        int a, b;
        heap.pop(&fa, &a);
        heap.pop(&fb, &b);
        parent[a] = next_id;
        parent[b] = next_id;
        height[static_cast<size_t>(next_id)] = static_cast<uint8_t>(std::max(height[static_cast<size_t>(a)], height[static_cast<size_t>(b)]) + 1);
        heap.push(fa + fb, next_id);
        ++next_id;
    }
    // code length = depth below the root; a parent's id is larger than its children's, so one pass from the root down
    std::vector<uint8_t> depth_of(static_cast<size_t>(next_id), 0);
    for (int id = next_id - 2; id >= 0; --id) depth_of[static_cast<size_t>(id)] = static_cast<uint8_t>(depth_of[static_cast<size_t>(parent[id])] + 1);
    // CanonicalSorter :27-35 orders by (length, symbol): leaves in symbol order (the pseudo-EOF is the largest symbol), then a
    // stable counting sort by length
    std::vector<int> leaf_of(table_size, -1);
    for (int l = 0; l < eof_leaf; ++l) leaf_of[symbols[l]] = l;
    std::vector<int> by_symbol;
    by_symbol.reserve(static_cast<size_t>(leaves));
    for (size_t v = 0; v < table_size; ++v)
        if (leaf_of[v] >= 0) by_symbol.push_back(leaf_of[v]);
    by_symbol.push_back(eof_leaf);
    auto length_of_leaf = [&](int l) { return (n == 0 && l == eof_leaf) ? uint8_t(1) : depth_of[static_cast<size_t>(l)]; };   // :70 the lone pseudo-EOF gets length 1
    size_t first_of_length[257] = {};
    for (int l : by_symbol) ++first_of_length[static_cast<size_t>(length_of_leaf(l)) + 1];
    for (int k = 1; k <= 256; ++k) first_of_length[k] += first_of_length[k - 1];
    std::vector<Entry> entries(static_cast<size_t>(leaves));
    for (int l : by_symbol) {
        const uint8_t len = length_of_leaf(l);
        entries[first_of_length[len]++] = Entry{symbols[l], len, 0};
    }
    const uint8_t max_length = std::max<uint8_t>(height[static_cast<size_t>(next_id - 1)], 1);      // the root is the last node made
    t.max_length = max_length;
    out.put(max_length, 8);
    std::vector<uint16_t> group_sizes;
    t.code_of.assign(table_size, 0);
    t.length_of.assign(table_size, 0);
    uint8_t prev_length = 0;
    uint32_t code = 0, count = 0;
    for (Entry& e : entries) {
        if (e.length != prev_length) {
            if (prev_length != 0) {
                out.put(static_cast<uint16_t>(count), 16);
                group_sizes.push_back(static_cast<uint16_t>(count));
            }
            for (uint8_t l = static_cast<uint8_t>(prev_length + 1); l < e.length; ++l) {
                out.put(0, 16);
                group_sizes.push_back(0);
            }
            count = 0;
            code <<= (e.length - prev_length);
        }
        e.code = code++;
        ++count;
        prev_length = e.length;
        if (e.symbol == kPseudoEof) { t.eof_code = e.code; t.eof_length = e.length; }
        else { t.code_of[e.symbol] = e.code; t.length_of[e.symbol] = e.length; }
    }
    out.put(static_cast<uint16_t>(count), 16);
    group_sizes.push_back(static_cast<uint16_t>(count));
    out.put(symbol_bits, 8);
    const uint16_t mask = static_cast<uint16_t>((1u << symbol_bits) - 1u);
    size_t at = 0;
    std::vector<uint16_t> group;
    for (size_t g = 0; at < entries.size(); ++g) {
        const uint16_t size = group_sizes[g];
        group.resize(size);
        for (uint16_t k = 0; k < size; ++k) group[k] = static_cast<uint16_t>(entries[at + k].symbol) & mask;
        if (elias_fano_length(size, mask) < static_cast<uint32_t>(size) * symbol_bits) elias_fano_write(group.data(), size, mask, out);
        else for (uint16_t v : group) out.put(v, symbol_bits);
        at += size;
    }
    t.payload_bits = 0;
    for (int l = 0; l < eof_leaf; ++l) t.payload_bits += freq[l] * static_cast<size_t>(t.length_of[symbols[l]]);
}

void huffman_encode_with(const SymbolStats& st, const uint16_t* data, size_t n, BitWriter& out) {
    HuffmanTable t;
    huffman_table(st, n, out, t);
    out.put_codes(data, n, t.code_of.data(), t.length_of.data(), t.payload_bits);
    out.put(t.eof_code, t.eof_length);
}
}  // namespace

// Huffman.cpp:173-244
bool huffman_decode(BitReader& in, std::vector<uint16_t>& out) {
    const int max_length = static_cast<int>(in.get(8));
    std::vector<uint16_t> counts(static_cast<size_t>(max_length));
    size_t entries = 0;
    for (int l = 0; l < max_length; ++l) {
        counts[l] = static_cast<uint16_t>(in.get(16));
        entries += counts[l];
    }
    // the reference keeps this sum in a uint16_t (Huffman.cpp:176): a table of more than 65535 entries wraps there and
    // cannot come from its encoder; a symbol width above 16 overflows its mask.  Both are invalid data here.
    if (entries > 65535) return false;
    const uint16_t total = static_cast<uint16_t>(entries);
    const int symbol_bits = static_cast<int>(in.get(8));
    if (symbol_bits < 1 || symbol_bits > 16) return false;
    if (entries > in.remaining()) return false;                 // every table entry takes at least one bit
    const uint16_t mask = static_cast<uint16_t>((1u << symbol_bits) - 1u);
    std::vector<uint16_t> table(total);
    size_t at = 0;
    for (int l = 0; l < max_length; ++l) {
        const uint16_t size = counts[l];
        if (elias_fano_length(size, mask) < static_cast<uint32_t>(size) * static_cast<uint32_t>(symbol_bits)) {
            if (!elias_fano_read(table.data() + at, size, mask, in)) return false;
        } else {
            for (uint16_t k = 0; k < size; ++k) table[at + k] = static_cast<uint16_t>(in.get(symbol_bits));
        }
        at += size;
    }
    if (total == 0 || table[total - 1u] != mask) return false;      // the last entry must be the pseudo-EOF
    std::vector<uint32_t> first_code(static_cast<size_t>(max_length) + 1, 0), first_index(static_cast<size_t>(max_length) + 1, 0);
    uint32_t code = 0, index = 0;
    int prev = 0;
    for (int l = 1; l <= max_length; ++l) {
        if (!counts[l - 1]) continue;
        code <<= (l - prev);
        first_code[l] = code;
        first_index[l] = index;
        code += counts[l - 1];
        index += counts[l - 1];
        prev = l;
    }
    // The reference shifts bits in one at a time and takes the first length at which the accumulated value is a code
    // of that length.  Same decisions here from a 32-bit look-ahead: codes of up to kLutBits bits through a table
    // (filled shortest length first, so an over-subscribed table still resolves to the first match), longer ones by
    // the reference's test per length.
    if (max_length > 32) {                                      // no encoder writes this; keep the literal bit-by-bit walk
        uint64_t acc = 0;
        int bits = 0;
        while (in.remaining() > 0) {
            acc = (acc << 1) | in.get(1);
            ++bits;
            if (bits <= 32 && counts[bits - 1] && acc >= first_code[bits] && acc - first_code[bits] < counts[bits - 1]) {
                const uint32_t entry = first_index[bits] + static_cast<uint32_t>(acc - first_code[bits]);
                if (entry == static_cast<uint32_t>(total) - 1u) return true;
                out.push_back(table[entry]);
                acc = 0;
                bits = 0;
            }
        }
        return false;
    }
    constexpr int kLutBits = 11;
    std::vector<uint32_t> lut(static_cast<size_t>(1) << kLutBits, 0);           // (entry << 5) | length, 0 = no short code
    for (int l = 1; l <= std::min(max_length, kLutBits); ++l) {
        for (uint32_t k = 0; k < counts[l - 1]; ++k) {
            const uint64_t value = static_cast<uint64_t>(first_code[l]) + k;
            if (value >> l) break;                                              // not an l-bit number: can never match
            const uint32_t lo = static_cast<uint32_t>(value) << (kLutBits - l);
            for (uint32_t fill = 0; fill < (1u << (kLutBits - l)); ++fill)
                if (lut[lo + fill] == 0) lut[lo + fill] = ((first_index[l] + k) << 5) | static_cast<uint32_t>(l);
        }
    }
    // two symbols per look-up where two whole codes (neither the pseudo-EOF) fit the window: same symbols in the same order
    std::vector<uint32_t> pair_symbols(lut.size(), 0);
    std::vector<uint8_t> pair_bits(lut.size(), 0);
    const uint32_t eof_entry = static_cast<uint32_t>(total) - 1u;
    for (uint32_t w = 0; w < lut.size(); ++w) {
        const uint32_t h1 = lut[w];
        if (h1 == 0 || (h1 >> 5) == eof_entry) continue;
        const uint32_t l1 = h1 & 31u;
        if (l1 >= static_cast<uint32_t>(kLutBits)) continue;
        const uint32_t h2 = lut[(w << l1) & (static_cast<uint32_t>(lut.size()) - 1u)];
        if (h2 == 0 || (h2 >> 5) == eof_entry || l1 + (h2 & 31u) > static_cast<uint32_t>(kLutBits)) continue;
        pair_symbols[w] = static_cast<uint32_t>(table[h1 >> 5]) | (static_cast<uint32_t>(table[h2 >> 5]) << 16);
        pair_bits[w] = static_cast<uint8_t>(l1 + (h2 & 31u));
    }
    // One table for the fast loop: symbols (2 x 16 bits), bits consumed (8), symbols decoded (0 = not here: a code longer than
    // the table's window, or the pseudo-EOF -- the careful loop below takes those, and everything within eight bytes of the end).
    std::vector<uint64_t> fast(lut.size(), 0);
    for (uint32_t w = 0; w < lut.size(); ++w) {
        if (pair_bits[w]) fast[w] = static_cast<uint64_t>(pair_symbols[w]) | (static_cast<uint64_t>(pair_bits[w]) << 32) | (2ULL << 40);
        else if (lut[w] != 0 && (lut[w] >> 5) != eof_entry)
            fast[w] = static_cast<uint64_t>(table[lut[w] >> 5]) | (static_cast<uint64_t>(lut[w] & 31u) << 32) | (1ULL << 40);
    }
    // Codes longer than the table's window, in the fast loop: when no length is over-subscribed (first_code[l] + counts[l-1] <=
    // 2^l for every l -- true of every table an encoder writes), the codes of the used lengths, left-aligned, tile [0, end) in
    // order of length, so "the first length at which the accumulated value is a code" is the one whose interval holds the next
    // bits: one comparison per used length.  Any other table takes the reference's test length by length.
    bool tiled = true;
    std::vector<uint64_t> end_left(static_cast<size_t>(max_length) + 1, 0);      // left-aligned (64-bit) end of length l's codes
    for (int l = 1; l <= max_length; ++l) {
        if (!counts[l - 1]) continue;
        const uint64_t end = static_cast<uint64_t>(first_code[l]) + counts[l - 1];
        if (end > (1ULL << l)) { tiled = false; break; }
        end_left[l] = end << (64 - l);                                           // end == 2^l wraps to 0: treated below
    }
    // ... and, for such tables, a second level behind the fast table: the window's entries that are prefixes of longer codes point
    // to a sub-table indexed by the next w bits (w = the longest code with that prefix - kLutBits), one look-up more instead of a
    // walk over the lengths.  (symbol | length << 16; length 0 = no code, 0xFF = the pseudo-EOF: both go to the careful loop.)
    std::vector<uint32_t> sub;
    if (tiled && max_length > kLutBits && max_length <= 26) {
        std::vector<uint8_t> wbits(lut.size(), 0);
        for (int l = kLutBits + 1; l <= max_length; ++l)
            for (uint32_t k = 0; k < counts[l - 1]; ++k) {
                const uint32_t prefix = (first_code[l] + k) >> (l - kLutBits);
                wbits[prefix] = std::max<uint8_t>(wbits[prefix], static_cast<uint8_t>(l - kLutBits));
            }
        std::vector<uint32_t> off(lut.size(), 0);
        size_t total_sub = 0;
        for (size_t pfx = 0; pfx < lut.size() && total_sub <= (1u << 16); ++pfx)
            if (wbits[pfx]) { off[pfx] = static_cast<uint32_t>(total_sub); total_sub += size_t(1) << wbits[pfx]; }
        if (total_sub <= (1u << 16)) {
            sub.assign(total_sub, 0);
            for (int l = kLutBits + 1; l <= max_length; ++l)
                for (uint32_t k = 0; k < counts[l - 1]; ++k) {
                    const uint32_t v = first_code[l] + k, x = static_cast<uint32_t>(l - kLutBits), prefix = v >> x, w = wbits[prefix];
                    const uint32_t entry = first_index[l] + k;
                    const uint32_t val = entry == eof_entry ? 0x00FF0000u : (static_cast<uint32_t>(table[entry]) | (static_cast<uint32_t>(l) << 16));
                    const uint32_t base = (v & ((1u << x) - 1u)) << (w - x);
                    for (uint32_t r = 0; r < (1u << (w - x)); ++r) sub[off[prefix] + base + r] = val;
                }
            for (size_t pfx = 0; pfx < lut.size(); ++pfx)
                if (wbits[pfx] && fast[pfx] == 0 && lut[pfx] == 0)
                    fast[pfx] = static_cast<uint64_t>(off[pfx]) | (static_cast<uint64_t>(wbits[pfx]) << 32) | (3ULL << 40);
        }
    }
    const uint8_t* const bytes = in.data();
    const size_t total_bits = in.size_bits();
    // the symbols go to out[o++]; `out` is kept larger than o (grown geometrically, cut to o on the way out)
    size_t o = out.size();
    struct Trim {
        std::vector<uint16_t>& v;
        size_t& n;
        ~Trim() { v.resize(n); }
    } trim{out, o};
    auto room = [&](size_t more) {
        if (out.size() < o + more) out.resize(std::max<size_t>(out.size() + out.size() / 2, o + more + 4096));
    };
    for (;;) {
        if (total_bits >= 128) {
            // A 64-bit bit buffer, refilled without branches: its top `cnt` bits are the next bits of the stream (bits below them
            // may already hold what follows: harmless); consumed so far = 8 * (ptr - bytes) - cnt.  Up to four table look-ups per
            // refill (at most kLutBits bits each); a code longer than the window is resolved from the same buffer.
            const size_t pos = in.position();
            const uint8_t* ptr = bytes + (pos >> 3);
            const uint8_t* const last = bytes + total_bits / 8 - 8;         // an 8-byte load at `last` still lies inside the data
            uint64_t buf = 0;
            unsigned cnt = 0;
            auto refill = [&] {
                uint64_t w;
                std::memcpy(&w, ptr, 8);
                buf |= __builtin_bswap64(w) >> cnt;
                ptr += (63 - cnt) >> 3;
                cnt |= 56;
            };
            bool careful = ptr > last;
            if (!careful) {
                refill();
                buf <<= (pos & 7);
                cnt -= static_cast<unsigned>(pos & 7);
            }
            while (!careful && ptr <= last) {
                room(8);
                refill();                                                   // cnt >= 56
                uint16_t* dst = out.data() + o;
                int k = 0;
                uint64_t e = 0;
                for (; k < 4; ++k) {
                    e = fast[buf >> (64 - kLutBits)];
                    if (((e >> 40) - 1) > 1) break;                          // neither one nor two symbols here
                    dst[0] = static_cast<uint16_t>(e);
                    dst[1] = static_cast<uint16_t>(e >> 16);
                    dst += (e >> 40);
                    const unsigned n = static_cast<unsigned>((e >> 32) & 0xFFu);
                    buf <<= n;
                    cnt -= n;
                }
                o = static_cast<size_t>(dst - out.data());
                if (k == 4) continue;
                if ((e >> 40) == 3) {                                       // a longer code, through its prefix's sub-table
                    const unsigned w = static_cast<unsigned>((e >> 32) & 0xFFu);
                    if (cnt < static_cast<unsigned>(kLutBits) + w) {
                        if (ptr > last) break;
                        refill();
                    }
                    const uint32_t e2 = sub[static_cast<uint32_t>(e) + static_cast<uint32_t>((buf << kLutBits) >> (64 - w))];
                    const unsigned len = e2 >> 16;
                    if (len == 0 || len == 0xFFu) { careful = true; break; }
                    out[o++] = static_cast<uint16_t>(e2);
                    buf <<= len;
                    cnt -= len;
                    continue;
                }
                // not in the table: the pseudo-EOF (careful loop), or a code longer than the window
                if (lut[buf >> (64 - kLutBits)] != 0) { careful = true; break; }
                if (cnt < static_cast<unsigned>(max_length)) {
                    if (ptr > last) break;
                    refill();
                }
                int l = kLutBits + 1;
                uint32_t acc = 0;
                if (tiled) {
                    for (; l <= max_length; ++l)
                        if (counts[l - 1] && (buf < end_left[l] || end_left[l] == 0)) break;
                    if (l <= max_length) {
                        acc = static_cast<uint32_t>(buf >> (64 - l));
                        if (acc < first_code[l]) l = max_length + 1;         // below every longer code: not a code at all
                    }
                } else {
                    for (; l <= max_length; ++l) {                           // the reference's test per length, on the buffer
                        acc = static_cast<uint32_t>(buf >> (64 - l));
                        if (counts[l - 1] && acc >= first_code[l] && acc - first_code[l] < counts[l - 1]) break;
                    }
                }
                if (l > max_length) { careful = true; break; }               // no code: the careful loop fails as the reference does
                const uint32_t entry = first_index[l] + (acc - first_code[l]);
                if (entry == eof_entry) { careful = true; break; }
                out[o++] = table[entry];
                buf <<= l;
                cnt -= static_cast<unsigned>(l);
            }
            if (ptr >= bytes + (pos >> 3) + 1 || cnt != 0) in.set_position(8 * static_cast<size_t>(ptr - bytes) - cnt);
        }
        // one symbol the careful way (a long code, the pseudo-EOF, the last bytes of the data)
        const size_t left = in.remaining();
        if (left == 0) return false;
        const uint32_t window = in.peek32();
        uint32_t entry = 0;
        int used = 0;
        const uint32_t hit = lut[window >> (32 - kLutBits)];
        if (hit != 0 && (hit & 31u) <= left) {
            entry = hit >> 5;
            used = static_cast<int>(hit & 31u);
        } else {
            for (int l = (hit != 0 ? 1 : kLutBits + 1); l <= max_length && static_cast<size_t>(l) <= left; ++l) {
                const uint32_t acc = window >> (32 - l);
                if (counts[l - 1] && acc >= first_code[l] && acc - first_code[l] < counts[l - 1]) {
                    entry = first_index[l] + (acc - first_code[l]);
                    used = l;
                    break;
                }
            }
            if (used == 0) return false;                        // the reference reads on to the end of the data and fails
        }
        in.skip(static_cast<size_t>(used));
        if (entry == static_cast<uint32_t>(total) - 1u) return true;
        room(1);
        out[o++] = table[entry];
    }
}

// Huffman.cpp:246-310
std::vector<uint16_t> rle_encode(const uint16_t* data, size_t n) {
    std::vector<uint16_t> out;
    out.reserve(n);
    uint16_t prev = 0, run = 0;
    bool fresh = true;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t v = data[i];
        if (v == prev && !fresh) {
            if (++run == 1) out.push_back(v);
            else if (run >= 0x8000) { out.push_back(static_cast<uint16_t>(run - 1)); run = 0; fresh = true; }
        } else {
            fresh = false;
            if (run > 0) { out.push_back(static_cast<uint16_t>(run - 1)); run = 0; }
            prev = v;
            out.push_back(v);
        }
    }
    if (run > 0) out.push_back(static_cast<uint16_t>(run - 1));
    return out;
}

std::vector<uint16_t> rle_decode(const uint16_t* data, size_t n) {
    std::vector<uint16_t> out;
    out.reserve(n);
    uint16_t prev = 0;
    bool expect_count = false, fresh = true;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t v = data[i];
        if (expect_count) {
            out.insert(out.end(), v, prev);
            expect_count = false;
            fresh = true;
        } else {
            out.push_back(v);
            if (v == prev && !fresh) expect_count = true;
            fresh = false;
            prev = v;
        }
    }
    return out;
}

// size rle_decode would produce, or false once it exceeds `limit`
bool rle_decoded_size(const uint16_t* data, size_t n, size_t limit, size_t* size) {
    size_t total = 0;
    uint16_t prev = 0;
    bool expect_count = false, fresh = true;
    for (size_t i = 0; i < n; ++i) {
        const uint16_t v = data[i];
        if (expect_count) {
            total += v;
            expect_count = false;
            fresh = true;
        } else {
            ++total;
            if (v == prev && !fresh) expect_count = true;
            fresh = false;
            prev = v;
        }
        if (total > limit) return false;
    }
    *size = total;
    return true;
}

// CompressedImage.cpp:359-401.  The Golomb cost of each candidate M is the sum over DISTINCT symbols of
// count x length -- same number as the reference's per-symbol loop.
void write_huffman_or_golomb(const uint16_t* data, size_t n, BitWriter& out) {
    const SymbolStats st = gather_stats(data, n);
    BitWriter huff;
    huffman_encode_with(st, data, n, huff);
    size_t best = huff.bit_size();
    int best_m = -1;
    for (int m = 1; m < 2048; m = (m & 1) ? m + 1 : (m << 1) - 1) {
        size_t estimate = 16;
        for (uint16_t s : st.distinct) estimate += static_cast<size_t>(st.hist[s]) * golomb_length(s, static_cast<uint32_t>(m));
        if (estimate < best) { best = estimate; best_m = m; }
    }
    if (best_m < 0) {
        out.put(0, 1);
        out.append(huff);
    } else {
        out.put(1, 1);
        out.put(static_cast<uint16_t>(best_m), 16);
        for (size_t i = 0; i < n; ++i) golomb_write(data[i], static_cast<uint32_t>(best_m), out);
    }
}

// writeHuffmanOrGolomb and the run-length wrapper of writeCompressed (:449-453) decided from a stream's statistics alone --
// the device has counted (mp_entropy.hip) and will write the codes; the host builds the table.  Same decisions, same bits:
// the Huffman cost is table + sum of count x length + pseudo-EOF, the Golomb cost as in write_huffman_or_golomb.
void plan_stream(bool rle_flag, bool shorter, uint32_t rle_size, size_t n, uint32_t largest, const uint32_t* triples, size_t distinct,
                 StreamPlan& plan) {
    SymbolStats st;
    st.largest = static_cast<uint16_t>(largest);
    st.hist.assign(n ? static_cast<size_t>(largest) + 1 : 1, 0);
    // order of first appearance (positions are distinct): position << 16 | symbol, sorted by position -- a radix sort, 11 bits
    // a pass, over as many passes as the largest position has bits (a comparison sort of a deltaId stream's ~1 700 symbols
    // took as long as everything else in here but the tree)
    std::vector<uint64_t> by_first(distinct), other(distinct);
    uint32_t last_position = 0;
    for (size_t d = 0; d < distinct; ++d) {
        const uint32_t symbol = triples[3 * d];
        st.hist[symbol] = triples[3 * d + 1];
        by_first[d] = (static_cast<uint64_t>(triples[3 * d + 2]) << 16) | symbol;
        last_position = std::max(last_position, triples[3 * d + 2]);
    }
    if (distinct < 128) {
        std::sort(by_first.begin(), by_first.end());
        last_position = 0;                                      // done
    }
    for (int shift = 16; shift < 48 && (static_cast<uint64_t>(last_position) << 16 >> shift) != 0; shift += 11) {
        uint32_t start[2049] = {};
        for (uint64_t v : by_first) ++start[((v >> shift) & 2047u) + 1];
        for (int b = 1; b <= 2048; ++b) start[b] += start[b - 1];
        for (uint64_t v : by_first) other[start[(v >> shift) & 2047u]++] = v;
        by_first.swap(other);
    }
    st.distinct.resize(distinct);
    for (size_t d = 0; d < distinct; ++d) st.distinct[d] = static_cast<uint16_t>(by_first[d] & 0xFFFFu);

    plan = StreamPlan();
    if (rle_flag) {
        plan.pre.put(shorter ? 1 : 0, 1);
        if (shorter) plan.pre.put(rle_size, 32);
    }
    BitWriter table_bits;
    HuffmanTable t;
    huffman_table(st, n, table_bits, t);
    size_t best = table_bits.bit_size() + t.payload_bits + t.eof_length;
    int best_m = -1;
    for (int m = 1; m < 2048; m = (m & 1) ? m + 1 : (m << 1) - 1) {
        size_t estimate = 16;
        for (uint16_t s : st.distinct) {                        // a candidate is out as soon as its partial sum reaches the best so far
            estimate += static_cast<size_t>(st.hist[s]) * golomb_length(s, static_cast<uint32_t>(m));
            if (estimate >= best) break;
        }
        if (estimate < best) { best = estimate; best_m = m; }
    }
    if (best_m < 0) {
        plan.pre.put(0, 1);
        plan.pre.append(table_bits);
        plan.post.put(t.eof_code, t.eof_length);
        plan.payload_bits = t.payload_bits;
        plan.mode = 0;
        plan.max_code_length = t.max_length;
        plan.entries.reserve(3 * distinct);
        for (uint16_t s : st.distinct) {
            plan.entries.push_back(s);
            plan.entries.push_back(t.code_of[s]);
            plan.entries.push_back(t.length_of[s]);
        }
    } else {
        plan.pre.put(1, 1);
        plan.pre.put(static_cast<uint16_t>(best_m), 16);
        plan.payload_bits = best - 16;
        plan.mode = 1;
        plan.m = static_cast<uint32_t>(best_m);
    }
}

void or_bits(uint8_t* dst, size_t dst_bytes, size_t bit_offset, const BitWriter& piece) {
    const size_t nbits = piece.bit_size();
    const uint64_t* src = piece.words();                        // MSB first, zero beyond nbits
    for (size_t done = 0; done < nbits; done += 64) {
        const uint64_t word = src[done >> 6];
        const size_t at = bit_offset + done, byte = at >> 3;
        const int sh = static_cast<int>(at & 7);
        if (byte + 9 <= dst_bytes) {                            // 64 bits at once: eight bytes and the spill into a ninth
            uint64_t have;
            std::memcpy(&have, dst + byte, 8);
            have |= __builtin_bswap64(word >> sh);
            std::memcpy(dst + byte, &have, 8);
            if (sh) dst[byte + 8] |= static_cast<uint8_t>((word << (64 - sh)) >> 56);
            continue;
        }
        const size_t take = std::min<size_t>(64, nbits - done);   // near the end of the buffer: bit ranges, byte by byte
        size_t pos = at;
        for (size_t k = 0; k < take;) {
            const int off = static_cast<int>(pos & 7);
            const int width = static_cast<int>(std::min<size_t>(8 - off, take - k));
            const uint8_t bits = static_cast<uint8_t>((word >> (64 - k - width)) & ((1u << width) - 1u));
            dst[pos >> 3] |= static_cast<uint8_t>(bits << (8 - off - width));
            pos += static_cast<size_t>(width);
            k += static_cast<size_t>(width);
        }
    }
}

bool read_huffman_or_golomb(BitReader& in, size_t length, std::vector<uint16_t>& out) {
    if (in.get(1) == 0) {
        out.reserve(out.size() + std::min(length, in.remaining()));      // every code takes at least one bit
        return huffman_decode(in, out);
    }
    const uint32_t m = static_cast<uint32_t>(in.get(16));
    if (m == 0) return false;
    if (length > in.remaining()) return false;                  // every Golomb code takes at least one bit: a lying header
    const size_t first = out.size();
    out.resize(first + length);
    uint16_t* dst = out.data() + first;
    size_t i = 0;
    {   // whole 64-bit words while eight bytes remain: unary part by counting leading ones, remainder from the same word
        const uint32_t b = bit_width(m), limit = (1u << (b + 1)) - m;
        const uint8_t* const bytes = in.data();
        const size_t total_bits = in.size_bits();
        if (total_bits >= 64) {
            const size_t safe = total_bits - 64;
            size_t pos = in.position();
            while (i < length && pos <= safe) {
                uint64_t w;
                std::memcpy(&w, bytes + (pos >> 3), 8);
                w = __builtin_bswap64(w) << (pos & 7);                      // at least 57 valid bits
                const unsigned ones = static_cast<unsigned>(__builtin_clzll(~w | 1ULL));
                if (ones + 1 + b + 1 > 57) break;                           // a long unary run: the careful reader
                const uint64_t rest = w << (ones + 1);
                const uint32_t head = b ? static_cast<uint32_t>(rest >> (64 - b)) : 0u;
                uint32_t rem = head;
                unsigned used = ones + 1 + b;
                if (head >= limit) {
                    rem = (head << 1) + static_cast<uint32_t>((rest >> (63 - b)) & 1ULL) - limit;
                    ++used;
                }
                dst[i++] = static_cast<uint16_t>(ones * m + rem);
                pos += used;
            }
            in.set_position(pos);
        }
    }
    for (; i < length; ++i) dst[i] = static_cast<uint16_t>(golomb_read(m, in));
    return true;
}

namespace {
constexpr uint32_t kMagic = 0x4D4E3234u;        // CompressedImage.cpp:14

std::vector<uint16_t> dc_difference(const std::vector<uint16_t>& v) {       // :428-446
    std::vector<uint16_t> out(v.size());
    int32_t prev = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        out[i] = static_cast<uint16_t>(zigzag_encode(static_cast<int32_t>(v[i]) - prev));
        prev = static_cast<int32_t>(v[i]);
    }
    return out;
}
}  // namespace

namespace {
// The 1 + 6K streams of a container are coded independently and only concatenated bit-wise afterwards, so
// they are coded on a small thread pool (the reference is single-threaded; the bytes do not depend on it).
int host_threads() {
    if (const char* v = std::getenv("MPC_HOST_THREADS")) {
        const int n = std::atoi(v);
        if (n > 0) return n;
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return static_cast<int>(hc == 0 ? 1 : (hc > 16 ? 16 : hc));
}

// A small persistent pool: the workers are created on first use and sleep between calls (creating and joining 16
// threads per call cost more than coding a 1080p frame's streams).  One parallel_for at a time (callers serialise on
// `submit_`); the calling thread works too.
class WorkerPool {
public:
    static WorkerPool& instance() {
        static WorkerPool pool;
        return pool;
    }
    static WorkerPool& io_instance() {                      // a second pool: frame uploads run beside the entropy stage's jobs
        static WorkerPool pool;
        return pool;
    }
    void run(int n, int workers, const std::function<void(int)>& body) {
        std::lock_guard<std::mutex> one_at_a_time(submit_);
        ensure(workers - 1);
        {
            std::lock_guard<std::mutex> hold(lock_);
            body_ = &body;
            total_ = n;
            next_.store(0);
            pending_ = std::min<int>(workers - 1, static_cast<int>(threads_.size()));
            active_limit_ = pending_;
            ++generation_;
        }
        wake_.notify_all();
        work(body, n);
        // Whatever a job threw, the generation is drained before `body` (the caller's stack) goes away; the first exception is
        // then rethrown on the calling thread, where the C ABI's `guarded` turns it into a status.
        std::exception_ptr first;
        {
            std::unique_lock<std::mutex> hold(lock_);
            done_.wait(hold, [&] { return pending_ == 0; });
            body_ = nullptr;
            first = error_;
            error_ = nullptr;
        }
        if (first) std::rethrow_exception(first);
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> hold(lock_);
            stop_ = true;
            ++generation_;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }

private:
    // this thread's share of the jobs; a job that throws ends the call's remaining jobs (nobody starts another) and is remembered
    void work(const std::function<void(int)>& body, int total) {
        try {
            for (int i = next_.fetch_add(1); i < total; i = next_.fetch_add(1)) body(i);
        } catch (...) {
            next_.store(total);
            std::lock_guard<std::mutex> hold(lock_);
            if (!error_) error_ = std::current_exception();
        }
    }
    void ensure(int count) {
        while (static_cast<int>(threads_.size()) < count) {
            const int id = static_cast<int>(threads_.size());
            threads_.emplace_back([this, id] { loop(id); });
        }
    }
    void loop(int id) {
        unsigned long long seen = 0;
        for (;;) {
            const std::function<void(int)>* body = nullptr;
            int total = 0;
            {
                std::unique_lock<std::mutex> hold(lock_);
                wake_.wait(hold, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                if (id >= active_limit_) continue;             // this call wants fewer workers
                body = body_;
                total = total_;
            }
            work(*body, total);
            {
                std::lock_guard<std::mutex> hold(lock_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex submit_, lock_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* body_ = nullptr;
    std::atomic<int> next_{0};
    int total_ = 0, pending_ = 0, active_limit_ = 0;
    unsigned long long generation_ = 0;
    bool stop_ = false;
    std::exception_ptr error_;
};

template <class F>
void parallel_for(int n, F&& body) {
    const int workers = std::min(host_threads(), n);
    if (workers <= 1) {
        for (int i = 0; i < n; ++i) body(i);
        return;
    }
    const std::function<void(int)> fn = [&](int i) { body(i); };
    WorkerPool::instance().run(n, workers, fn);
}
}  // namespace

BitWriter container_head(int width, int height, int K, int block_size, const double* quant) {
    BitWriter head;
    head.put(kMagic, 32);
    head.put(static_cast<uint32_t>(width), 32);
    head.put(static_cast<uint32_t>(height), 32);
    head.put(static_cast<uint8_t>(K), 8);
    head.put(static_cast<uint8_t>(block_size), 8);
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) head.put(static_cast<uint16_t>(quant[ch * K + i]), 16);    // :420 u16 of an integral double
    return head;
}

void parallel_jobs(int n, const std::function<void(int)>& body) { parallel_for(n, body); }

void parallel_io_jobs(int n, int workers, const std::function<void(int)>& body) {
    workers = std::min(std::min(workers, host_threads()), n);
    if (workers <= 1) {
        for (int i = 0; i < n; ++i) body(i);
        return;
    }
    WorkerPool::io_instance().run(n, workers, body);
}

std::vector<uint8_t> write_compressed(const Streams& s) {
    const int K = s.K;
    BitWriter out;
    out.put(kMagic, 32);
    out.put(static_cast<uint32_t>(s.width), 32);
    out.put(static_cast<uint32_t>(s.height), 32);
    out.put(static_cast<uint8_t>(K), 8);
    out.put(static_cast<uint8_t>(s.block_size), 8);
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) out.put(s.quant[ch][i], 16);
    std::vector<BitWriter> parts(static_cast<size_t>(6 * K + 1));
    parallel_for(6 * K + 1, [&](int job) {
        BitWriter& w = parts[static_cast<size_t>(job)];
        if (job == 0) {
            write_huffman_or_golomb(s.lengths.data(), s.lengths.size(), w);
            return;
        }
        const int i = job - 1;
        const bool dc = (i == 1 || i == 2 * K + 1 || i == 4 * K + 1);
        const std::vector<uint16_t> diffed = dc ? dc_difference(s.codes[i]) : std::vector<uint16_t>();
        const std::vector<uint16_t>& stream = dc ? diffed : s.codes[i];
        const bool shorter = rle_encoded_size(stream.data(), stream.size()) + 4 < stream.size();
        const std::vector<uint16_t> packed = shorter ? rle_encode(stream.data(), stream.size()) : std::vector<uint16_t>();
        if (shorter) {                                           // :450
            w.put(1, 1);
            w.put(static_cast<uint32_t>(packed.size()), 32);
            write_huffman_or_golomb(packed.data(), packed.size(), w);
        } else {
            w.put(0, 1);
            write_huffman_or_golomb(stream.data(), stream.size(), w);
        }
    });
    for (const BitWriter& w : parts) out.append(w);
    return out.bytes();
}

// assemble_streams + write_compressed in one go, without materialising the 6K streams of a frame (hundreds of MB of
// freshly faulted pages at K = 32): one job per (channel, step) gathers its two streams into buffers the worker thread
// keeps between calls and codes them straight into its part of the container.  Same bytes as the two-step route.
namespace {
// record (tile t, channel ch, step i) = choices[t * tile_stride + ch * channel_stride + i * step_stride]
struct RecordLayout {
    size_t tile_stride, channel_stride, step_stride;
};

void code_records(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts, const uint32_t* choices,
                  const RecordLayout& layout, BitWriter& out, std::vector<BitWriter>& parts) {
    const size_t tiles = static_cast<size_t>((width + block_size - 1) / block_size) *
                         static_cast<size_t>((height + block_size - 1) / block_size);
    out.put(kMagic, 32);
    out.put(static_cast<uint32_t>(width), 32);
    out.put(static_cast<uint32_t>(height), 32);
    out.put(static_cast<uint8_t>(K), 8);
    out.put(static_cast<uint8_t>(block_size), 8);
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) out.put(static_cast<uint16_t>(quant[ch * K + i]), 16);    // :420 u16 of an integral double
    parts.assign(static_cast<size_t>(6 * K + 1), BitWriter());
    auto code_stream = [](const std::vector<uint16_t>& stream, bool dc, std::vector<uint16_t>& scratch, BitWriter& w) {
        const std::vector<uint16_t>* src = &stream;
        if (dc) {
            scratch = dc_difference(stream);
            src = &scratch;
        }
        const bool shorter = rle_encoded_size(src->data(), src->size()) + 4 < src->size();
        const std::vector<uint16_t> packed = shorter ? rle_encode(src->data(), src->size()) : std::vector<uint16_t>();
        if (shorter) {                                          // :450
            w.put(1, 1);
            w.put(static_cast<uint32_t>(packed.size()), 32);
            write_huffman_or_golomb(packed.data(), packed.size(), w);
        } else {
            w.put(0, 1);
            write_huffman_or_golomb(src->data(), src->size(), w);
        }
    };
    // the big jobs first: step 0 of every channel holds every tile-channel, later steps fewer
    parallel_for(3 * K + 1, [&](int job0) {
        if (job0 == 0) {
            write_huffman_or_golomb(counts, 3 * tiles, parts[0]);                              // the lengths stream: the longest
            return;
        }
        const int job = job0 - 1;
        const int i = job / 3, ch = job - 3 * i;                 // job order: (step 0: Y U V), (step 1: Y U V), ...
        thread_local std::vector<uint16_t> d, c, scratch;
        const uint32_t* mine = choices + static_cast<size_t>(ch) * layout.channel_stride + static_cast<size_t>(i) * layout.step_stride;
        const size_t tile_stride = layout.tile_stride;
        d.clear();
        c.clear();
        for (size_t t = 0; t < tiles; ++t) {
            if (counts[3 * t + static_cast<size_t>(ch)] > i) {
                const uint32_t rec = mine[t * tile_stride];
                d.push_back(static_cast<uint16_t>(rec & 0xFFFFu));
                c.push_back(static_cast<uint16_t>(rec >> 16));
            }
        }
        const int index = 2 * K * ch + 2 * i;                    // codes[index] = deltaId, [index + 1] = intCoeff
        code_stream(d, false, scratch, parts[static_cast<size_t>(index + 1)]);
        code_stream(c, i == 0, scratch, parts[static_cast<size_t>(index + 2)]);   // DC: the step-0 coefficients (:428-446)
    });
}

// Concatenate head and parts bit-wise into big-endian bytes in a malloc'ed buffer.  Every part knows its bit offset, so
// the parts are shifted into place in parallel; only the two words a part may share with its neighbours are merged
// with atomic ORs (into words cleared beforehand), everything in between is a plain store.
uint8_t* concat_malloc(const BitWriter& head, const std::vector<BitWriter>& parts, size_t* nbytes) {
    std::vector<size_t> offset(parts.size() + 1);
    size_t total = head.bit_size();
    for (size_t p = 0; p < parts.size(); ++p) {
        offset[p] = total;
        total += parts[p].bit_size();
    }
    offset[parts.size()] = total;
    const size_t nwords = (total + 63) / 64;
    uint64_t* dst = static_cast<uint64_t*>(std::malloc((nwords ? nwords : 1) * sizeof(uint64_t)));
    if (!dst) return nullptr;
    auto place = [dst](const BitWriter& w, size_t bit_offset) {
        const size_t nbits = w.bit_size();
        if (nbits == 0) return;
        const uint64_t* src = w.words();
        const size_t src_words = (nbits + 63) / 64;
        const size_t w0 = bit_offset >> 6, last = (bit_offset + nbits - 1) >> 6;
        const int shift = static_cast<int>(bit_offset & 63);
        for (size_t d = w0; d <= last; ++d) {                   // destination word d = source bits [64(d-w0) - shift, +64)
            const size_t i = d - w0;
            uint64_t v = 0;
            if (shift == 0) v = i < src_words ? src[i] : 0;
            else {
                if (i < src_words) v |= src[i] >> shift;
                if (i >= 1 && i - 1 < src_words) v |= src[i - 1] << (64 - shift);
            }
            const uint64_t be = __builtin_bswap64(v);            // MSB-first bit order = big-endian bytes
            if (d == w0 || d == last) __atomic_fetch_or(&dst[d], be, __ATOMIC_RELAXED);
            else dst[d] = be;
        }
    };
    // clear the words that can be shared between neighbours (first and last word of every piece)
    auto clear_ends = [dst](size_t bit_offset, size_t nbits) {
        if (nbits == 0) return;
        dst[bit_offset >> 6] = 0;
        dst[(bit_offset + nbits - 1) >> 6] = 0;
    };
    clear_ends(0, head.bit_size());
    for (size_t p = 0; p < parts.size(); ++p) clear_ends(offset[p], parts[p].bit_size());
    place(head, 0);
    parallel_for(static_cast<int>(parts.size()), [&](int p) { place(parts[static_cast<size_t>(p)], offset[static_cast<size_t>(p)]); });
    *nbytes = (total + 7) / 8;
    return reinterpret_cast<uint8_t*>(dst);
}
}  // namespace

std::vector<uint8_t> encode_records(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                    const uint32_t* choices) {
    BitWriter out;
    std::vector<BitWriter> parts;
    code_records(width, height, K, block_size, quant, counts, choices, RecordLayout{3 * static_cast<size_t>(K), static_cast<size_t>(K), 1}, out, parts);
    for (const BitWriter& w : parts) out.append(w);
    return out.bytes();
}

uint8_t* encode_records_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                               const uint32_t* choices, size_t* nbytes) {
    BitWriter head;
    std::vector<BitWriter> parts;
    code_records(width, height, K, block_size, quant, counts, choices, RecordLayout{3 * static_cast<size_t>(K), static_cast<size_t>(K), 1}, head, parts);
    return concat_malloc(head, parts, nbytes);
}

// The container from streams the device has already assembled (mp_streams.hip): `symbols` holds codes[0], codes[1], ... codes[6K-1]
// back to back (stream s = symbols[off[s] .. off[s+1])), live symbols only, in the reference's tile order, the three step-0
// coefficient streams already difference coded.  One job per stream (the lengths stream first: the longest).
uint8_t* encode_symbol_streams_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                      const uint16_t* symbols, const unsigned long long* off, size_t* nbytes) {
    const size_t tiles = static_cast<size_t>((width + block_size - 1) / block_size) *
                         static_cast<size_t>((height + block_size - 1) / block_size);
    BitWriter head;
    head.put(kMagic, 32);
    head.put(static_cast<uint32_t>(width), 32);
    head.put(static_cast<uint32_t>(height), 32);
    head.put(static_cast<uint8_t>(K), 8);
    head.put(static_cast<uint8_t>(block_size), 8);
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) head.put(static_cast<uint16_t>(quant[ch * K + i]), 16);    // :420 u16 of an integral double
    std::vector<BitWriter> parts(static_cast<size_t>(6 * K + 1));
    // longest jobs first: the lengths stream, then the streams in the order of their sizes
    std::vector<int> order(static_cast<size_t>(6 * K));
    for (int s = 0; s < 6 * K; ++s) order[static_cast<size_t>(s)] = s;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return off[x + 1] - off[x] > off[y + 1] - off[y]; });
    parallel_for(6 * K + 1, [&](int job) {
        if (job == 0) {
            write_huffman_or_golomb(counts, 3 * tiles, parts[0]);
            return;
        }
        const int s = order[static_cast<size_t>(job - 1)];
        const uint16_t* data = symbols + off[s];
        const size_t n = static_cast<size_t>(off[s + 1] - off[s]);
        BitWriter& w = parts[static_cast<size_t>(s + 1)];
        const bool shorter = rle_encoded_size(data, n) + 4 < n;
        const std::vector<uint16_t> packed = shorter ? rle_encode(data, n) : std::vector<uint16_t>();
        if (shorter) {                                            // CompressedImage.cpp:450
            w.put(1, 1);
            w.put(static_cast<uint32_t>(packed.size()), 32);
            write_huffman_or_golomb(packed.data(), packed.size(), w);
        } else {
            w.put(0, 1);
            write_huffman_or_golomb(data, n, w);
        }
    });
    return concat_malloc(head, parts, nbytes);
}

// encode_symbol_streams_malloc by the route the device-side entropy stage takes, with the device's share done here on the
// host: per-stream statistics -> plan_stream -> codes at the planned bit offsets -> OR the pieces into place.  Exists so that
// the planning half can be checked against the direct route without a GPU (tests/test_host_bitstream.py).
uint8_t* encode_symbol_streams_by_plan_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                              const uint16_t* symbols, const unsigned long long* off, size_t* nbytes) {
    const size_t tiles = static_cast<size_t>((width + block_size - 1) / block_size) *
                         static_cast<size_t>((height + block_size - 1) / block_size);
    const int S = 6 * K + 1;
    std::vector<StreamPlan> plans(static_cast<size_t>(S));
    std::vector<BitWriter> payload(static_cast<size_t>(S));
    for (int j = 0; j < S; ++j) {
        const uint16_t* data = j == 0 ? counts : symbols + off[j - 1];
        const size_t n = j == 0 ? 3 * tiles : static_cast<size_t>(off[j] - off[j - 1]);
        const size_t rle_size = j == 0 ? n : rle_encoded_size(data, n);
        const bool shorter = j != 0 && rle_size + 4 < n;
        const std::vector<uint16_t> packed = shorter ? rle_encode(data, n) : std::vector<uint16_t>();
        const uint16_t* coded = shorter ? packed.data() : data;
        const size_t coded_n = shorter ? packed.size() : n;
        std::vector<uint32_t> hist(65536, 0), first(65536, 0), triples;
        uint32_t largest = 0;
        for (size_t i = 0; i < coded_n; ++i) {
            if (hist[coded[i]]++ == 0) first[coded[i]] = static_cast<uint32_t>(i);
            largest = std::max<uint32_t>(largest, coded[i]);
        }
        for (uint32_t v = 0; v < 65536; ++v)
            if (hist[v]) { triples.push_back(v); triples.push_back(hist[v]); triples.push_back(first[v]); }
        StreamPlan& p = plans[static_cast<size_t>(j)];
        plan_stream(j != 0, shorter, static_cast<uint32_t>(rle_size), coded_n, largest, triples.data(), triples.size() / 3, p);
        BitWriter& w = payload[static_cast<size_t>(j)];
        if (p.mode == 0) {
            std::vector<uint32_t> code_of(static_cast<size_t>(largest) + 1, 0);
            std::vector<uint8_t> length_of(static_cast<size_t>(largest) + 1, 0);
            for (size_t k = 0; k < p.entries.size(); k += 3) {
                code_of[p.entries[k]] = p.entries[k + 1];
                length_of[p.entries[k]] = static_cast<uint8_t>(p.entries[k + 2]);
            }
            w.put_codes(coded, coded_n, code_of.data(), length_of.data(), p.payload_bits);
        } else {
            for (size_t i = 0; i < coded_n; ++i) golomb_write(coded[i], p.m, w);
        }
        if (w.bit_size() != p.payload_bits) return nullptr;
    }
    const BitWriter head = container_head(width, height, K, block_size, quant);
    size_t total = head.bit_size();
    for (int j = 0; j < S; ++j) total += plans[static_cast<size_t>(j)].pre.bit_size() + plans[static_cast<size_t>(j)].payload_bits + plans[static_cast<size_t>(j)].post.bit_size();
    *nbytes = (total + 7) / 8;
    uint8_t* dst = static_cast<uint8_t*>(std::calloc(*nbytes ? *nbytes : 1, 1));
    if (!dst) return nullptr;
    or_bits(dst, *nbytes, 0, head);
    size_t at = head.bit_size();
    for (int j = 0; j < S; ++j) {
        const StreamPlan& p = plans[static_cast<size_t>(j)];
        or_bits(dst, *nbytes, at, p.pre);
        at += p.pre.bit_size();
        or_bits(dst, *nbytes, at, payload[static_cast<size_t>(j)]);
        at += p.payload_bits;
        or_bits(dst, *nbytes, at, p.post);
        at += p.post.bit_size();
    }
    return dst;
}

uint8_t* encode_planar_records_malloc(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                      const uint32_t* planar, size_t* nbytes) {
    const size_t tiles = static_cast<size_t>((width + block_size - 1) / block_size) *
                         static_cast<size_t>((height + block_size - 1) / block_size);
    BitWriter head;
    std::vector<BitWriter> parts;
    code_records(width, height, K, block_size, quant, counts, planar, RecordLayout{1, static_cast<size_t>(K) * tiles, tiles}, head, parts);
    return concat_malloc(head, parts, nbytes);
}


bool read_compressed(const uint8_t* bytes, size_t nbytes, Streams& s) {
    BitReader in(bytes, nbytes);
    if (static_cast<uint32_t>(in.get(32)) != kMagic) return false;
    s.width = static_cast<int>(in.get(32));
    s.height = static_cast<int>(in.get(32));
    s.K = static_cast<int>(in.get(8));
    s.block_size = static_cast<int>(in.get(8));
    if (s.K < 1 || s.K > 32 || s.block_size < 1 || s.block_size > 8 || s.width < 1 || s.height < 1) return false;
    const int K = s.K;
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) s.quant[ch][i] = static_cast<uint16_t>(in.get(16));
    const size_t tiles = static_cast<size_t>((s.width + s.block_size - 1) / s.block_size) *
                         static_cast<size_t>((s.height + s.block_size - 1) / s.block_size);
    s.lengths.clear();
    if (tiles > (static_cast<size_t>(1) << 40) / 3) return false;
    if (!read_huffman_or_golomb(in, 3 * tiles, s.lengths)) return false;
    // a Huffman-coded lengths stream carries its own end: it must still describe exactly this frame's tiles (the
    // device decoder walks tiles_x * tiles_y records)
    if (s.lengths.size() != 3 * tiles) return false;
    s.codes.assign(static_cast<size_t>(6 * K), {});
    // length of an un-packed stream = tile-channels of its layer with more than `depth` atoms (:680-685): suffix sums
    // of the histogram of lengths, once for all 6K streams
    std::vector<size_t> expect_of(static_cast<size_t>(3 * K), 0);
    {
        std::vector<size_t> hist(static_cast<size_t>(3) * 65536, 0);
        for (size_t t = 0; t < s.lengths.size() / 3; ++t)
            for (size_t layer = 0; layer < 3; ++layer) ++hist[layer * 65536 + s.lengths[3 * t + layer]];
        for (size_t layer = 0; layer < 3; ++layer) {
            size_t above = 0;
            for (int v = 65535; v > K; --v) above += hist[layer * 65536 + static_cast<size_t>(v)];
            for (int depth = K - 1; depth >= 0; --depth) {
                above += hist[layer * 65536 + static_cast<size_t>(depth + 1)];
                expect_of[layer * static_cast<size_t>(K) + static_cast<size_t>(depth)] = above;
            }
        }
    }
    // The codes are self-delimiting, so the bit stream is parsed serially -- but only the entropy codes: run-length expansion
    // (:660-678) and the DC sums (:690-705) of a stream need nothing from the streams behind it and run on the pool afterwards.
    std::vector<std::vector<uint16_t>> packed(static_cast<size_t>(6 * K));
    std::vector<char> is_packed(static_cast<size_t>(6 * K), 0);
    std::vector<size_t> expect(static_cast<size_t>(6 * K), 0);
    for (int i = 0; i < 6 * K; ++i) {
        expect[i] = expect_of[(static_cast<size_t>(i / 2) / K) * static_cast<size_t>(K) + static_cast<size_t>(i / 2) % K];
        if (in.get(1) == 1) {
            const size_t packed_len = static_cast<size_t>(in.get(32));
            is_packed[i] = 1;
            if (!read_huffman_or_golomb(in, packed_len, packed[i])) return false;
        } else {
            if (!read_huffman_or_golomb(in, expect[i], s.codes[i])) return false;
        }
    }
    std::vector<char> bad(static_cast<size_t>(6 * K), 0);
    parallel_for(6 * K, [&](int i) {
        if (is_packed[i]) {
            // run lengths come from the data: refuse to expand beyond what the lengths stream allows for this stream
            size_t expanded = 0;
            if (!rle_decoded_size(packed[i].data(), packed[i].size(), expect[i], &expanded)) { bad[i] = 1; return; }
            s.codes[i] = rle_decode(packed[i].data(), packed[i].size());
        }
        if (s.codes[i].size() != expect[i]) { bad[i] = 1; return; }
        if (i == 1 || i == 2 * K + 1 || i == 4 * K + 1) {       // :690-705
            int32_t acc = 0;
            for (uint16_t& c : s.codes[i]) {
                acc += zigzag_decode(c);
                c = static_cast<uint16_t>(acc);
            }
        }
    });
    for (int i = 0; i < 6 * K; ++i)
        if (bad[i]) return false;
    return true;
}

Streams assemble_streams(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                         const uint32_t* choices) {
    Streams s;
    s.width = width;
    s.height = height;
    s.K = K;
    s.block_size = block_size;
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) s.quant[ch][i] = static_cast<uint16_t>(quant[ch * K + i]);   // :420 u16 of an integral double
    const size_t tiles = static_cast<size_t>((width + block_size - 1) / block_size) *
                         static_cast<size_t>((height + block_size - 1) / block_size);
    s.lengths.resize(3 * tiles);
    s.codes.assign(static_cast<size_t>(6 * K), {});
    // stream sizes first, then fill: no reallocation on multi-megabyte streams.  Step i of a channel holds one symbol
    // per tile-channel with count > i: suffix sums of the histogram of counts.
    std::vector<size_t> sizes(static_cast<size_t>(3 * K), 0);
    {
        std::vector<size_t> hist(static_cast<size_t>(3 * (K + 1)), 0);
        for (size_t o = 0; o < 3 * tiles; ++o) {
            s.lengths[o] = counts[o];
            ++hist[(o % 3) * static_cast<size_t>(K + 1) + std::min<size_t>(counts[o], static_cast<size_t>(K))];
        }
        for (int ch = 0; ch < 3; ++ch) {
            size_t above = 0;
            for (int i = K - 1; i >= 0; --i) {
                above += hist[static_cast<size_t>(ch) * (K + 1) + static_cast<size_t>(i + 1)];
                sizes[static_cast<size_t>(ch * K + i)] = above;
            }
        }
    }
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) {
            s.codes[2 * K * ch + 2 * i].reserve(sizes[ch * K + i]);
            s.codes[2 * K * ch + 2 * i + 1].reserve(sizes[ch * K + i]);
        }
    // one job per (channel, step): each fills its own pair of streams by scanning that channel's counts
    parallel_for(3 * K, [&](int job) {
        const int ch = job / K, i = job - ch * K;
        std::vector<uint16_t>& d = s.codes[2 * K * ch + 2 * i];
        std::vector<uint16_t>& c = s.codes[2 * K * ch + 2 * i + 1];
        const uint32_t* mine = choices + static_cast<size_t>(ch) * K + static_cast<size_t>(i);
        const size_t tile_stride = 3 * static_cast<size_t>(K);
        for (size_t t = 0; t < tiles; ++t) {
            if (counts[3 * t + static_cast<size_t>(ch)] > i) {
                const uint32_t rec = mine[t * tile_stride];
                d.push_back(static_cast<uint16_t>(rec & 0xFFFFu));
                c.push_back(static_cast<uint16_t>(rec >> 16));
            }
        }
    });
    return s;
}

bool disassemble_streams(const Streams& s, uint16_t* counts, uint32_t* choices) {
    const int K = s.K;
    const size_t n = s.lengths.size(), tiles = n / 3;
    if (n % 3 != 0) return false;
    for (int i = 0; i < 6 * K; i += 2)
        if (s.codes[i].size() != s.codes[i + 1].size()) return false;
    // Blocks of tiles in parallel: where a block starts in each of the 3K stream pairs = tile-channels of the blocks in front of
    // it with more than `step` atoms (suffix sums of a histogram of the block's counts, then a running sum over the blocks).
    const size_t block = 4096, blocks = (tiles + block - 1) / block;
    std::vector<size_t> start((blocks + 1) * static_cast<size_t>(3 * K), 0);
    std::vector<char> bad(blocks, 0);
    parallel_for(static_cast<int>(blocks), [&](int b) {
        std::vector<size_t> hist(static_cast<size_t>(3 * (K + 1)), 0);
        const size_t lo = block * static_cast<size_t>(b), hi = std::min(tiles, lo + block);
        for (size_t o = 3 * lo; o < 3 * hi; ++o) {
            if (s.lengths[o] > K) { bad[b] = 1; return; }
            ++hist[(o % 3) * static_cast<size_t>(K + 1) + s.lengths[o]];
        }
        size_t* mine = start.data() + (static_cast<size_t>(b) + 1) * static_cast<size_t>(3 * K);
        for (int ch = 0; ch < 3; ++ch) {
            size_t above = 0;
            for (int i = K - 1; i >= 0; --i) {
                above += hist[static_cast<size_t>(ch) * (K + 1) + static_cast<size_t>(i + 1)];
                mine[ch * K + i] = above;
            }
        }
    });
    for (size_t b = 0; b < blocks; ++b)
        if (bad[b]) return false;
    for (size_t b = 1; b <= blocks; ++b)
        for (int p = 0; p < 3 * K; ++p) start[b * static_cast<size_t>(3 * K) + p] += start[(b - 1) * static_cast<size_t>(3 * K) + p];
    for (int p = 0; p < 3 * K; ++p)                                 // every stream must hold what the lengths promise
        if (start[blocks * static_cast<size_t>(3 * K) + p] > s.codes[2 * p].size()) return false;
    parallel_for(static_cast<int>(blocks), [&](int b) {
        std::vector<size_t> cursor(start.begin() + static_cast<size_t>(b) * (3 * K), start.begin() + (static_cast<size_t>(b) + 1) * (3 * K));
        const size_t lo = block * static_cast<size_t>(b), hi = std::min(tiles, lo + block);
        for (size_t o = 3 * lo; o < 3 * hi; ++o) {
            const int ch = static_cast<int>(o % 3);
            const int count = s.lengths[o];
            counts[o] = s.lengths[o];
            for (int i = 0; i < count; ++i) {
                const size_t at = cursor[static_cast<size_t>(ch * K + i)]++;
                choices[o * K + i] = static_cast<uint32_t>(s.codes[2 * K * ch + 2 * i][at]) | (static_cast<uint32_t>(s.codes[2 * K * ch + 2 * i + 1][at]) << 16);
            }
        }
    });
    return true;
}

bool disassemble_streams(const Streams& s, std::vector<uint16_t>& counts, std::vector<uint32_t>& choices) {
    counts.assign(s.lengths.size(), 0);
    choices.assign(s.lengths.size() * static_cast<size_t>(s.K), 0);
    return disassemble_streams(s, counts.data(), choices.data());
}

}  // namespace mpc
