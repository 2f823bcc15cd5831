// mp_entropy.hip -- product: the per-symbol work of the entropy stage on the device (SURVEY 8f N2:
// "RLE, Golomb/Huffman bit-length costing from histograms", CompressedImage.cpp:359-458).
//
// The container's 1 + 6K streams (`lengths`, then codes[0..6K)) are coded independently: run-length coded if that is
// shorter (:449-453, Huffman.cpp:246-279), then Huffman or Golomb, whichever is shorter (:359-390).  Everything that
// touches every symbol runs here; the host keeps what is sequential and small -- one Huffman tree per stream, built from
// the stream's histogram and the order in which its symbols first appear (that order decides the reference's ties, see
// host_bitstream.cpp: MsvcHashOrder) -- and hands back one code table per stream.
//
//   phase 1 (after stream assembly, same stream)
//     layout   sizes and block tables of the 1 + 6K streams (from the stream offsets, which only the device knows)
//     runs     per block of 4096 symbols: where runs of equal symbols start and end; symbols runLengthEncode emits for
//              every run that starts inside the block
//     plan     per stream: the run a block continues (a segmented scan over the blocks), symbols emitted for those in
//              closed form, output offsets, the reference's decision `packed.size() + 4 < stream.size()`
//     pack     the run-length coded stream, for the streams where it is shorter
//     hist     histogram and first position of every symbol of the stream that will be coded, in LDS (a workgroup per span of
//              the stream and range of 8192 symbol values), and the number of distinct symbols (a bin's first count)
//     compact  (symbol, count, first position) of the symbols that occur, per stream, in one list written straight to host memory
//     mirror   the per-stream records to host memory
//   host: Huffman tables / Golomb parameter, bit offsets of every stream's payload in the container
//   phase 2
//     tables   the host's part of the per-stream records in; scatter the (symbol -> code, length) entries into dense per-stream tables
//     count    code bits per block;  offsets  exclusive scan per stream, from the stream's bit offset
//     write    the codes, MSB first, into the zeroed container (32-bit atomic ORs of byte-swapped words)
//     clear    the table entries, for the next frame; the per-stream records (bit counts) back to host memory
// Integer work, HBM/L2-bound; 2 bytes per symbol and pass, ~10 M symbols per 16 Mpixel frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mp_device.h"

namespace mpc {

namespace {
constexpr int kThreads = 256;
constexpr int kPer = kEntBlock / kThreads;              // 16 consecutive symbols per thread
constexpr unsigned kChunk = 0x8001u;                    // a run is cut after this many symbols (Huffman.cpp:262-267)
constexpr int kLdsBins = 8192;
constexpr unsigned kNoPos = 0xFFFFFFFFu;
static_assert(kPer * kThreads == kEntBlock, "block shape");

__device__ __forceinline__ const uint16_t* raw_stream(const EntropyArgs& a, int j, const EntStream& s)
{
    return j == 0 ? a.counts : a.symbols + s.raw_off;
}

// stream of scan block `b` (the layout kernel's table: a search over the streams' first blocks would be eight dependent loads
// in front of every workgroup's work)
__device__ __forceinline__ int find_stream(const EntropyArgs& a, unsigned b) { return (int)a.blk_stream[b]; }

__device__ __forceinline__ unsigned wave_incl_add(unsigned v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// exclusive prefix sum over the 256 threads of a block; *total = block sum.  `scratch`: 4 words of LDS.
__device__ __forceinline__ unsigned block_excl_add(unsigned v, unsigned* scratch, unsigned* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned incl = wave_incl_add(v);
    __syncthreads();                                     // scratch may still be read from an earlier call
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; ++w) {
        const unsigned t = scratch[w];
        if (w < wave) before += t;
        all += t;
    }
    *total = all;
    return before + incl - v;
}

// A block's symbols go through LDS: the threads of the runs and code kernels each own 16 CONSECUTIVE symbols (32 bytes -- a lane
// stride that makes every 2-byte load of a wave touch 64 separate lines), so the block is fetched by coalesced 4-byte loads
// and the threads pick their symbols out of LDS.  8 words are stored as 9: with the threads 8 words apart all lanes would
// share 4 banks.
constexpr int kStageWords = ((kEntBlock + 4 + 1) / 2 + 7) / 8 * 9;

__device__ __forceinline__ int staged_at(int s) { return s + ((s >> 4) << 1); }      // index in halves, padding included

// symbols [first, first + count) of the stream at `src` -> lds; the staged index of symbol `first` (0 or 1: the loads start at
// a 4-byte boundary; the word behind the last symbol may reach 2 bytes past it -- inside the buffers, which end 256-aligned)
__device__ __forceinline__ int stage_symbols(const uint16_t* src, unsigned first, unsigned count, uint32_t* lds)
{
    const uintptr_t addr = reinterpret_cast<uintptr_t>(src + first);
    const int shift = (int)((addr >> 1) & 1);
    const uint32_t* words = reinterpret_cast<const uint32_t*>(addr - 2 * (uintptr_t)shift);
    const int n_words = (int)((count + (unsigned)shift + 1u) >> 1);
    constexpr int kRounds = ((kEntBlock + 4 + 1) / 2 + kThreads - 1) / kThreads;      // all loads in flight before the first store
    uint32_t got[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int w = threadIdx.x + r * kThreads;
        got[r] = w < n_words ? words[w] : 0u;
    }
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int w = threadIdx.x + r * kThreads;
        if (w < n_words) lds[w + (w >> 3)] = got[r];
    }
    return shift;
}

// symbols runLengthEncode emits at offset c of a chunk, when the symbol there is / is not the last of its maximal run
__device__ __forceinline__ unsigned rle_emitted(unsigned c, bool run_end)
{
    return (c == 0) + (c == 1) + ((c >= 1) && (run_end || c == kChunk - 1));
}

// number of p in [0, x) with p mod kChunk == r
__device__ __forceinline__ unsigned residues_below(unsigned x, unsigned r) { return (x + kChunk - 1 - r) / kChunk; }

// symbols emitted for offsets [p0, p0 + len) of one run, the last of which ends the run or not
__device__ __forceinline__ unsigned rle_emitted_span(unsigned p0, unsigned len, bool run_end)
{
    if (len == 0) return 0;
    const unsigned hi = p0 + len;
    unsigned e = (residues_below(hi, 0) - residues_below(p0, 0)) + (residues_below(hi, 1) - residues_below(p0, 1)) +
                 (residues_below(hi, kChunk - 1) - residues_below(p0, kChunk - 1));
    const unsigned c = (hi - 1) % kChunk;
    if (run_end && c >= 1 && c != kChunk - 1) ++e;
    return e;
}
}  // namespace

// ---- phase 1 ----
__global__ __launch_bounds__(kThreads) void ent_layout_kernel(const EntropyArgs a)
{
    __shared__ unsigned nblk[kEntMaxStreams], nblk_end;
    const int j = threadIdx.x;
    if (j < a.n_streams) {
        const unsigned long long begin = j == 0 ? 0ULL : a.stream_off[j - 1];
        const unsigned n = j == 0 ? a.n_lengths : (unsigned)(a.stream_off[j] - begin);
        EntStream s{};
        s.raw_off = begin;
        s.n = n;
        s.eff_n = n;
        a.streams[j] = s;
        nblk[j] = (n + kEntBlock - 1) / kEntBlock;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned b = 0;
        for (int s = 0; s < a.n_streams; ++s) {
            a.streams[s].blk_begin = b;
            const unsigned n = nblk[s];
            nblk[s] = b;                                  // count -> first block
            b += n;
        }
        nblk_end = b;
        a.totals[0] = b;
        a.totals[1] = 0;
        a.totals[2] = 0;                                  // triples written
        a.totals[3] = 0;                                  // overflow flag
    }
    __syncthreads();
    if (j < a.n_streams) {
        const unsigned end = j + 1 < a.n_streams ? nblk[j + 1] : nblk_end;
        for (unsigned b = nblk[j]; b < end; ++b) a.blk_stream[b] = (unsigned)j;
    }
}

// kPack = false: per-block run structure;  kPack = true: write the run-length coded stream (offsets from the plan)
template <bool kPack>
__global__ __launch_bounds__(kThreads) void ent_runs_kernel(const EntropyArgs a)
{
    __shared__ unsigned scratch[kThreads / 64];
    __shared__ int first_bnd, last_bnd;
    __shared__ unsigned inner_sum;
    const unsigned b = blockIdx.x;
    if (b >= a.totals[0]) return;
    const int j = find_stream(a, b);
    const EntStream s = a.streams[j];
    const unsigned lb = b - s.blk_begin;
    const unsigned begin = lb * kEntBlock, len = min((unsigned)kEntBlock, s.n - begin);
    const uint16_t* src = raw_stream(a, j, s);
    if (j == 0) {                                         // `lengths` is never run-length coded (CompressedImage.cpp:424)
        if (!kPack) {
            unsigned largest = 0;
            for (unsigned i = threadIdx.x; i < len; i += kThreads) largest = max(largest, (unsigned)src[begin + i]);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) largest = max(largest, (unsigned)__shfl_xor((int)largest, d));
            if ((threadIdx.x & 63) == 0) atomicMax(&a.streams[0].largest, largest);
            if (threadIdx.x == 0) {
                a.blk_lead[b] = 0;
                a.blk_inner[b] = 0;
                a.blk_tail[b] = 0;
            }
        }
        return;
    }
    if (kPack && !s.shorter) return;
    if (threadIdx.x == 0) {
        first_bnd = (int)len;
        last_bnd = -1;
        inner_sum = 0;
    }
    // the thread's symbols with one neighbour on either side
    __shared__ uint32_t staged[kStageWords];
    const unsigned first_staged = begin ? begin - 1 : 0u;
    const int shift = stage_symbols(src, first_staged, min(s.n, begin + len + 1) - first_staged, staged);
    __syncthreads();
    const uint16_t* staged16 = reinterpret_cast<const uint16_t*>(staged);
    const unsigned t0 = threadIdx.x * kPer;
    uint16_t v[kPer + 2];
#pragma unroll
    for (int k = 0; k < kPer + 2; ++k) {
        const long long gi = (long long)begin + t0 + k - 1;
        v[k] = (gi >= 0 && gi < (long long)s.n && t0 + k <= len + 1) ? staged16[staged_at((int)(gi - first_staged) + shift)] : (uint16_t)0;
    }
    unsigned bnd_mask = 0, end_mask = 0;                  // bit k: position t0 + k starts / ends a maximal run
    int my_last = -1;
    unsigned largest = 0;                                 // of the symbols that will be coded (kPack: of the run lengths)
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        const unsigned i = t0 + k, gi = begin + i;
        if (i < len) {
            if (gi == 0 || v[k + 1] != v[k]) { bnd_mask |= 1u << k; my_last = (int)i; }
            if (gi == s.n - 1 || v[k + 2] != v[k + 1]) end_mask |= 1u << k;
            if (!kPack) largest = max(largest, (unsigned)v[k + 1]);
        }
    }
    __syncthreads();
    if (bnd_mask) {
        atomicMin(&first_bnd, (int)(t0 + (unsigned)__builtin_ctz(bnd_mask)));
        atomicMax(&last_bnd, my_last);
    }
    // start of the run that reaches into this thread's range: the last boundary of any earlier thread (max scan)
    int start = my_last;
    {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(start, d);
            if (lane >= d) start = max(start, o);
        }
        __shared__ int wave_last[kThreads / 64];
        if (lane == 63) wave_last[wave] = start;
        int excl = __shfl_up(start, 1);
        if (lane == 0) excl = -1;
        __syncthreads();
        for (int w = 0; w < wave; ++w) excl = max(excl, wave_last[w]);
        start = excl;
    }
    const unsigned carry = kPack ? a.blk_carry[b] : 0u;   // symbols of the run in front of the block (lead positions only)
    unsigned emitted = 0;
    unsigned e_of[kPer];
    {
        int cur = start;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const unsigned i = t0 + k;
            e_of[k] = 0;
            if (i < len) {
                if (bnd_mask & (1u << k)) cur = (int)i;
                if (cur >= 0 || kPack) {
                    const unsigned p = cur >= 0 ? i - (unsigned)cur : carry + i;
                    e_of[k] = rle_emitted(p % kChunk, (end_mask >> k) & 1u);
                    emitted += e_of[k];
                }
            }
        }
    }
    if (!kPack) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) largest = max(largest, (unsigned)__shfl_xor((int)largest, d));
        if ((threadIdx.x & 63) == 0) atomicMax(&a.streams[j].largest, largest);
        const unsigned w = wave_incl_add(emitted);
        if ((threadIdx.x & 63) == 63 && w) atomicAdd(&inner_sum, w);
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned lead = (unsigned)first_bnd;    // positions in front of the block's first boundary
            unsigned lead_end = 0;
            if (lead > 0) {
                const unsigned gi = begin + lead - 1;
                lead_end = (gi == s.n - 1) || (gi + 1 < s.n && src[gi + 1] != src[gi]);
            }
            a.blk_lead[b] = lead | (lead_end << 31);
            a.blk_inner[b] = inner_sum;
            a.blk_tail[b] = last_bnd >= 0 ? len - (unsigned)last_bnd : 0u;
        }
        return;
    }
    unsigned total;
    unsigned at = a.blk_out[b] + block_excl_add(emitted, scratch, &total);
    uint16_t* dst = a.packed + s.raw_off;
    {
        int cur = start;
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            const unsigned i = t0 + k;
            if (i < len) {
                if (bnd_mask & (1u << k)) cur = (int)i;
                const unsigned p = cur >= 0 ? i - (unsigned)cur : carry + i;
                const unsigned c = p % kChunk;
                if (c <= 1) dst[at++] = v[k + 1];
                if (c >= 1 && (((end_mask >> k) & 1u) || c == kChunk - 1)) {
                    dst[at++] = (uint16_t)(c - 1);
                    largest = max(largest, c - 1);
                }
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) largest = max(largest, (unsigned)__shfl_xor((int)largest, d));
    if ((threadIdx.x & 63) == 0 && largest > s.largest) atomicMax(&a.streams[j].largest, largest);
}

// one wave per stream
__global__ __launch_bounds__(64) void ent_rle_plan_kernel(const EntropyArgs a)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    EntStream s = a.streams[j];
    const unsigned nb = (s.n + kEntBlock - 1) / kEntBlock;
    unsigned run_before = 0;                              // length of the run that ends with the last symbol before this chunk
    unsigned out_before = 0;
    for (unsigned base = 0; base < nb; base += 64) {
        const unsigned lb = base + lane;
        const bool in = lb < nb;
        const unsigned b = s.blk_begin + (in ? lb : 0);
        const unsigned len = in ? min((unsigned)kEntBlock, s.n - lb * kEntBlock) : 0u;
        const unsigned lead_word = in ? a.blk_lead[b] : 0u;
        const unsigned lead = lead_word & 0x7FFFFFFFu;
        const bool lead_end = lead_word >> 31;
        const unsigned inner = in ? a.blk_inner[b] : 0u;
        const unsigned tail = in ? a.blk_tail[b] : 0u;
        // tail_run[b] = a boundary inside the block ? tail : tail_run[b - 1] + len   (segmented inclusive scan)
        unsigned val = (lead < len) ? tail : len;
        bool reset = (lead < len) || !in;
        if (!in) val = 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned ov = __shfl_up(val, d);
            const int orr = __shfl_up((int)reset, d);
            if (lane >= d && !reset) { val += ov; reset = orr != 0; }
        }
        if (!reset) val += run_before;                    // no boundary since the start of this chunk of blocks
        unsigned prev_run = __shfl_up(val, 1);
        if (lane == 0) prev_run = run_before;
        const unsigned carry = lead > 0 ? prev_run : 0u;
        const unsigned emit = inner + rle_emitted_span(carry, lead, lead_end);
        const unsigned incl = wave_incl_add(in ? emit : 0u);
        if (in) {
            a.blk_carry[b] = carry;
            a.blk_out[b] = out_before + incl - emit;
        }
        out_before += __shfl(incl, 63);
        run_before = __shfl(val, (int)min(63u, nb - 1 - base));
    }
    if (lane == 0) {
        s.rle_size = out_before;
        s.shorter = (j != 0 && (unsigned long long)out_before + 4ULL < s.n) ? 1u : 0u;      // CompressedImage.cpp:450
        s.eff_n = s.shorter ? out_before : s.n;
        a.streams[j] = s;
    }
}

// Histogram and first positions of the stream that will be coded.  A workgroup counts the symbols of one RANGE of kLdsBins
// values in one span of its stream, all in LDS: the deltaId streams use most of the 16-bit range (a few thousand distinct symbols
// each), and counting those with atomics on memory -- two per symbol -- cost more than everything else in this file together.
// A stream whose largest symbol needs r ranges gets r workgroups per span and spans r times as long, so the bins set up and
// flushed per symbol stay the same.  The grid's layout (which workgroup does what) follows from the streams' sizes and largest
// symbols, known only now: every workgroup derives it again (one scan over <= 193 streams).
__global__ __launch_bounds__(kThreads) void ent_hist_kernel(const EntropyArgs a)
{
    __shared__ unsigned hist[kLdsBins], first[kLdsBins];
    __shared__ unsigned wg_begin[kEntMaxStreams + 1];
    __shared__ unsigned scratch[kThreads / 64];
    static_assert(kEntMaxStreams <= kThreads, "one thread per stream in the layout scan");
    unsigned mine = 0;
    if ((int)threadIdx.x < a.n_streams) {
        const unsigned n = a.streams[threadIdx.x].eff_n, ranges = a.streams[threadIdx.x].largest / kLdsBins + 1;
        const unsigned span = kEntHistSpan * kEntBlock * ranges;
        mine = n ? ((n + span - 1) / span) * ranges : 0u;
    }
    unsigned total;
    const unsigned before = block_excl_add(mine, scratch, &total);
    if ((int)threadIdx.x < a.n_streams) wg_begin[threadIdx.x] = before;
    __syncthreads();
    const unsigned hb = blockIdx.x;
    if (hb >= total) return;
    int j = 0;
    {
        int lo = 0, hi = a.n_streams - 1;                 // the last stream with wg_begin <= hb that has workgroups at all
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (wg_begin[mid] <= hb) lo = mid; else hi = mid - 1;
        }
        j = lo;                                           // streams without workgroups share their successor's wg_begin: the
    }                                                     // search ends on the last of a run, which is the one with workgroups
    const EntStream s = a.streams[j];
    const unsigned ranges = s.largest / kLdsBins + 1, span = kEntHistSpan * kEntBlock * ranges;
    const unsigned local = hb - wg_begin[j], range = local % ranges;
    const unsigned begin = (local / ranges) * span;
    if (begin >= s.eff_n) return;
    const unsigned end = min(s.eff_n, begin + span);
    const uint16_t* src = s.shorter ? a.packed + s.raw_off : raw_stream(a, j, s);
    const unsigned lo_sym = range * kLdsBins;
    const int bins = (int)min((unsigned)kLdsBins, s.largest + 1 - lo_sym);       // the runs kernels have found the largest symbol
    for (int i = threadIdx.x; i < bins; i += kThreads) {
        hist[i] = 0;
        first[i] = kNoPos;
    }
    __syncthreads();
    constexpr int kInFlight = 8;                          // loads per thread before the first is used: two workgroups fit a CU
    for (unsigned base = begin + threadIdx.x; base < end; base += kInFlight * kThreads) {
        unsigned sym[kInFlight];
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const unsigned i = base + u * kThreads;
            sym[u] = i < end ? (unsigned)src[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < kInFlight; ++u) {
            const unsigned i = base + u * kThreads;
            const unsigned bin = sym[u] - lo_sym;         // below the range: wraps to something huge
            if (bin < (unsigned)bins) {
                atomicAdd(&hist[bin], 1u);
                if (first[bin] > i) atomicMin(&first[bin], i);                   // a bin's first position settles early; later symbols only look
            }
        }
    }
    __syncthreads();
    unsigned* ghist = a.ghist + (size_t)j * 65536 + lo_sym;
    unsigned* gfirst = a.gfirst + (size_t)j * 65536 + lo_sym;
    unsigned fresh = 0;                                   // bins this thread was the first to count into
    for (int i = threadIdx.x; i < bins; i += kThreads) {
        const unsigned h = hist[i];
        if (h) {
            fresh += atomicAdd(&ghist[i], h) == 0;
            atomicMin(&gfirst[i], first[i]);
        }
    }
    fresh = wave_incl_add(fresh);
    if ((threadIdx.x & 63) == 63 && fresh) atomicAdd(&a.streams[j].distinct, fresh);
}

// where each stream's triples go in the list: exclusive scan of the distinct counts (and the write cursors, cleared)
__global__ __launch_bounds__(kThreads) void ent_triple_offsets_kernel(const EntropyArgs a)
{
    __shared__ unsigned distinct[kEntMaxStreams];
    const int j = threadIdx.x;
    if (j < a.n_streams) {
        distinct[j] = a.streams[j].distinct;
        a.streams[j].reserved = 0;                        // the stream's write cursor
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned at = 0;
        for (int s = 0; s < a.n_streams; ++s) {
            a.streams[s].triple_off = at;
            at += distinct[s];
        }
        a.totals[2] = at;
        a.totals[3] = at > a.triple_cap;
    }
}

// grid (1024-bin chunk, stream): the bins that occur -> (symbol, count, first position) in the stream's part of the list (any
// order: the host sorts them by first position anyway); bins reset for the next frame
__global__ __launch_bounds__(kThreads) void ent_compact_kernel(const EntropyArgs a)
{
    const int j = blockIdx.y;
    const EntStream s = a.streams[j];
    const unsigned chunk = blockIdx.x * 4 * kThreads;
    if (s.eff_n == 0 || chunk > s.largest) return;
    unsigned* ghist = a.ghist + (size_t)j * 65536;
    unsigned* gfirst = a.gfirst + (size_t)j * 65536;
    const bool room = a.totals[3] == 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ULL << lane) - 1ULL;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned i = chunk + k * kThreads + threadIdx.x;
        const unsigned h = i <= s.largest ? ghist[i] : 0u;
        const unsigned long long hit = __ballot(h != 0);
        if (!hit) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&a.streams[j].reserved, (unsigned)__popcll(hit));
        base = __shfl(base, 0);
        if (h) {
            if (room) {
                unsigned* t = a.triples + 3 * (size_t)(s.triple_off + base + (unsigned)__popcll(hit & below));
                t[0] = i;
                t[1] = h;
                t[2] = gfirst[i];
            }
            ghist[i] = 0;
            gfirst[i] = kNoPos;
        }
    }
}

// The per-stream records and the totals, to their mirrors in host memory (and back, with the host's part filled in): kernels
// write them across PCIe themselves -- a copy command would be a blit kernel or an SDMA packet queued somewhere else, and this
// way the host needs nothing but the event behind the kernel.
__device__ __forceinline__ void mirror_streams(const EntropyArgs& a, int to_host)
{
    const int words = (int)(sizeof(EntStream) / sizeof(unsigned)) * a.n_streams;
    const unsigned* src = reinterpret_cast<const unsigned*>(to_host ? a.streams : a.host_streams);
    unsigned* dst = reinterpret_cast<unsigned*>(to_host ? a.host_streams : a.streams);
    for (int i = threadIdx.x; i < words; i += kThreads) dst[i] = src[i];
    if (to_host && threadIdx.x < 4) a.host_totals[threadIdx.x] = a.totals[threadIdx.x];
}

__global__ __launch_bounds__(kThreads) void ent_mirror_kernel(const EntropyArgs a, int to_host) { mirror_streams(a, to_host); }

// ---- phase 2 ----
// clear == 0: the host's records in (one extra workgroup, the last), (symbol -> code, length) entries into the dense tables;
// clear != 0: the entries out of the tables again, the records back to the host (their bit counts are final by then)
__global__ __launch_bounds__(kThreads) void ent_tables_kernel(const EntropyArgs a, int clear)
{
    if (blockIdx.x == gridDim.x - 1) {
        mirror_streams(a, clear ? 1 : 0);
        return;
    }
    const unsigned e = blockIdx.x * kThreads + threadIdx.x;
    if (e >= a.n_entries) return;
    const unsigned key = a.entries[3 * (size_t)e];        // stream << 16 | symbol
    if (clear) {
        a.tcode[key] = 0;
        a.tlen[key] = 0;
    } else {
        a.tcode[key] = a.entries[3 * (size_t)e + 1];
        a.tlen[key] = (uint8_t)a.entries[3 * (size_t)e + 2];
    }
}

namespace {
__device__ __forceinline__ unsigned bit_width_u32(unsigned v) { return v ? 32u - (unsigned)__builtin_clz(v) : 0u; }

// BitBuffer.cpp:228-269, the reference's own Golomb variant
__device__ __forceinline__ unsigned golomb_bits(unsigned value, unsigned m, unsigned b, unsigned limit)
{
    const unsigned q = value / m, rem = value - q * m;
    return b + q + 1 + (rem < limit ? 0u : 1u);
}
}  // namespace

// kWrite = false: code bits per block;  kWrite = true: the codes themselves
template <bool kWrite>
__global__ __launch_bounds__(kThreads) void ent_code_kernel(const EntropyArgs a)
{
    __shared__ unsigned scratch[kThreads / 64];
    const unsigned b = blockIdx.x;
    if (b >= a.totals[0]) return;
    const int j = find_stream(a, b);
    const EntStream s = a.streams[j];
    const unsigned begin = (b - s.blk_begin) * kEntBlock;
    if (begin >= s.eff_n) {
        if (!kWrite && threadIdx.x == 0) a.blk_bits[b] = 0;
        return;
    }
    const unsigned len = min((unsigned)kEntBlock, s.eff_n - begin);
    const uint16_t* src = (s.shorter ? a.packed + s.raw_off : raw_stream(a, j, s)) + begin;
    const unsigned t0 = threadIdx.x * kPer;
    const bool golomb = s.mode != 0;
    const unsigned m = golomb ? s.m : 1u;
    const unsigned gb = bit_width_u32(m), glimit = (1u << (gb + 1)) - m;
    const unsigned* tcode = a.tcode + (size_t)j * 65536;
    const uint8_t* tlen = a.tlen + (size_t)j * 65536;
    __shared__ uint32_t staged[kStageWords];
    const int shift = stage_symbols(src, 0, len, staged);
    __syncthreads();
    const uint16_t* staged16 = reinterpret_cast<const uint16_t*>(staged);
    unsigned sym[kPer], bits = 0;
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        sym[k] = 0;
        if (t0 + k < len) {
            sym[k] = staged16[staged_at((int)(t0 + k) + shift)];
            bits += golomb ? golomb_bits(sym[k], m, gb, glimit) : (unsigned)tlen[sym[k]];
        }
    }
    unsigned total;
    const unsigned before = block_excl_add(bits, scratch, &total);
    if (!kWrite) {
        if (threadIdx.x == 0) a.blk_bits[b] = total;
        return;
    }
    if (bits == 0) return;
    // MSB-first bit position P lives in byte P >> 3; a 32-bit big-endian word is OR-ed in byte-swapped
    unsigned long long pos = a.blk_bit_off[b] + before;
    unsigned long long w = pos >> 5;
    unsigned fill = (unsigned)(pos & 31), acc = 0;
    auto flush = [&]() {
        if (acc && w < a.out_words) atomicOr(a.out32 + w, __builtin_bswap32(acc));
    };
    auto append = [&](unsigned code, unsigned n) {        // n in 1..32, code < 2^n
        const unsigned room = 32 - fill;
        if (n < room) {
            acc |= code << (room - n);
            fill += n;
        } else {
            const unsigned rest = n - room;
            acc |= code >> rest;
            flush();
            ++w;
            acc = rest ? code << (32 - rest) : 0u;
            fill = rest;
        }
    };
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
        if (t0 + k < len) {
            if (!golomb) {
                const unsigned n = tlen[sym[k]];
                if (n) append(tcode[sym[k]], n);
            } else {
                unsigned q = sym[k] / m;
                const unsigned rem = sym[k] - q * m;
                while (q >= 32) { append(0xFFFFFFFFu, 32); q -= 32; }
                append(((1u << q) - 1u) << 1, q + 1);     // q ones and the terminating zero
                if (rem < glimit) { if (gb) append(rem, gb); }
                else append(rem + glimit, gb + 1);
            }
        }
    }
    flush();
}

// one wave per stream: bit offset of every block's codes
__global__ __launch_bounds__(64) void ent_bit_offsets_kernel(const EntropyArgs a)
{
    const int j = blockIdx.x, lane = threadIdx.x;
    const EntStream s = a.streams[j];
    const unsigned nb = (s.eff_n + kEntBlock - 1) / kEntBlock;
    unsigned long long at = s.bit_off;
    for (unsigned base = 0; base < nb; base += 64) {
        const unsigned lb = base + lane;
        const unsigned bits = lb < nb ? a.blk_bits[s.blk_begin + lb] : 0u;
        unsigned long long incl = bits;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lb < nb) a.blk_bit_off[s.blk_begin + lb] = at + incl - bits;
        at += __shfl(incl, 63);
    }
    if (lane == 0) a.streams[j].coded_bits = at - s.bit_off;
}

size_t entropy_max_blocks(unsigned long long capacity_symbols, int n_streams)
{
    return (size_t)((capacity_symbols + kEntBlock - 1) / kEntBlock) + (size_t)n_streams;
}

int launch_entropy_phase1(const EntropyArgs& a, unsigned long long capacity_symbols, void* stream_)
{
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (a.n_streams < 1 || a.n_streams > kEntMaxStreams) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)entropy_max_blocks(capacity_symbols, a.n_streams);
    // a stream of n symbols and r ranges: ceil(n / (span * r)) * r <= n / span + r workgroups, r <= 65536 / kLdsBins
    const unsigned hblocks = (unsigned)(capacity_symbols / (kEntHistSpan * kEntBlock)) + (unsigned)a.n_streams * (65536 / kLdsBins + 1);
    hipLaunchKernelGGL(ent_layout_kernel, dim3(1), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_runs_kernel<false>, dim3(blocks), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_rle_plan_kernel, dim3((unsigned)a.n_streams), dim3(64), 0, st, a);
    hipLaunchKernelGGL(ent_runs_kernel<true>, dim3(blocks), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_hist_kernel, dim3(hblocks), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_triple_offsets_kernel, dim3(1), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_compact_kernel, dim3(65536 / (4 * kThreads), (unsigned)a.n_streams), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_mirror_kernel, dim3(1), dim3(kThreads), 0, st, a, 1);
    return (int)hipGetLastError();
}

int launch_entropy_phase2(const EntropyArgs& a, unsigned long long raw_symbols, void* stream_)
{
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if (a.n_streams < 1 || a.n_streams > kEntMaxStreams) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)entropy_max_blocks(raw_symbols, a.n_streams);
    const unsigned eblocks = (a.n_entries + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(ent_tables_kernel, dim3(eblocks + 1), dim3(kThreads), 0, st, a, 0);
    hipLaunchKernelGGL(ent_code_kernel<false>, dim3(blocks), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_bit_offsets_kernel, dim3((unsigned)a.n_streams), dim3(64), 0, st, a);
    hipLaunchKernelGGL(ent_code_kernel<true>, dim3(blocks), dim3(kThreads), 0, st, a);
    hipLaunchKernelGGL(ent_tables_kernel, dim3(eblocks + 1), dim3(kThreads), 0, st, a, 1);
    return (int)hipGetLastError();
}

}  // namespace mpc
