// mpcodec_capi.cpp -- product: the C ABI of include/mpcodec.h over the host
// dictionary builder and the gfx950 kernels.  There is NO CPU fallback for the
// hot path: without a HIP device mpc_encode_tiles* return MPC_ERR_NO_DEVICE.
#include "../../include/mpcodec.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <utility>
#include <stdexcept>
#include <future>
#include <memory>
#include <mutex>
#include <vector>

#include "host_bitstream.h"
#include "mpc_internal.h"
#include "host_codec.h"
#include "host_stats.h"
#include "host_dictionary.h"
#include "mp_device.h"

namespace {

thread_local char g_error[512] = "";

mpc_status fail(mpc_status st, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
    return st;
}

#define HIP_TRY(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(MPC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

// No exception crosses the C ABI (the reference throws heap-allocated std::range_error*; here: status codes)
template <class F>
mpc_status guarded(F&& body) {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return fail(MPC_ERR_ALLOC, "out of memory");
    } catch (const std::length_error& e) {
        return fail(MPC_ERR_ALLOC, "allocation size out of range: %s", e.what());
    } catch (const std::exception& e) {
        return fail(MPC_ERR_BITSTREAM, "%s", e.what());
    } catch (...) {
        return fail(MPC_ERR_BITSTREAM, "unknown failure");
    }
}

template <class T>
hipError_t upload(T** dst, const T* src, size_t count) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dst), count * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice);
}

}  // namespace

// Everything on the device that depends only on (device, block size): the dictionary in double, its filter copies, the
// Gram table and the persistent kernel's scratch, streams and queues.  Shared by every context of the process on that
// device (a context adds K and the quantisation tables): the Gram table alone is 12.6 GB.  All persistent-kernel launches
// of the process go through the three per-channel streams held here, which also serialises their use of the scratch.
struct DeviceDict {
    int device = -1;
    int base_rows_padded = 0, num_base = 0, num_cus = 0;
    long long detail_rows = 0;
    double* d_base = nullptr;
    double* d_detail = nullptr;
    float* d_base32 = nullptr;        // the same rows rounded to float (the `...Fast` flavour), same layout
    float* d_detail32 = nullptr;
    int32_t* d_rows = nullptr;
    int32_t* d_rowoff = nullptr;
    uint16_t* d_base_t1 = nullptr;    // split-bf16 filter copies in MFMA operand order (persistent kernel)
    uint16_t* d_detail_t1 = nullptr;
    uint8_t* d_shadow = nullptr;      // [3][detail_rows]
    float* d_gram = nullptr;          // [3][num_base + detail_rows][num_base * 64]
    std::mutex launch_lock;           // one enqueue sequence at a time
    hipEvent_t done[1] = {};          // recorded behind every launch: the next one (any stream) waits for it
    int workgroups = 0;               // scratch is sized for this many workgroups per launch
    float* pair_p = nullptr;          // per-wave scratch of the persistent kernel (pairs: approximations, meta, bounds)
    unsigned* pair_meta = nullptr;
    float* pair_e = nullptr;
    unsigned* queues = nullptr;       // [3]
    unsigned long long* stats = nullptr;   // [2]: MFMA instructions, tile-channel-steps executed by the persistent kernel since the last reset
    ~DeviceDict() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        (void)hipFree(d_base); (void)hipFree(d_rows); (void)hipFree(d_rowoff);      // d_detail / d_detail32: inside d_base / d_base32
        (void)hipFree(d_base32);
        (void)hipFree(d_base_t1); (void)hipFree(d_detail_t1);
        (void)hipFree(d_shadow); (void)hipFree(d_gram); (void)hipFree(queues); (void)hipFree(stats);
        (void)hipFree(pair_p); (void)hipFree(pair_meta); (void)hipFree(pair_e);
        if (done[0]) (void)hipEventDestroy(done[0]);
    }
};

namespace {
std::mutex g_dicts_lock;
std::weak_ptr<DeviceDict> g_dicts[64];

// build (or share) the device residents of `dict` on `device`
hipError_t acquire_device_dict(int device, const mpc::Dictionary& dict, std::shared_ptr<DeviceDict>* out) {
    std::lock_guard<std::mutex> hold(g_dicts_lock);
    if (device < 64)
        if (std::shared_ptr<DeviceDict> have = g_dicts[device].lock()) { *out = have; return hipSuccess; }
    std::shared_ptr<DeviceDict> d = std::make_shared<DeviceDict>();
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    d->device = device;
    d->num_base = dict.num_base;
    d->detail_rows = dict.total_detail_rows();
    std::vector<double> base = mpc::base_padded(dict, 2, &d->base_rows_padded);
    const size_t det_rows = static_cast<size_t>(dict.total_detail_rows());
    // one zero row after the last: the exhaustive sweep's scalar prefetch reads one row past the rows it correlates
    std::vector<double> det((3 * det_rows + 1) * mpc::kTileN, 0.0);
    for (int ch = 0; ch < 3; ++ch)
        std::memcpy(det.data() + ch * det_rows * mpc::kTileN, dict.detail[ch].data(), det_rows * mpc::kTileN * sizeof(double));
    {   // base rows and detail rows in one allocation
        std::vector<double> all(base);
        all.insert(all.end(), det.begin(), det.end());
        e = upload(&d->d_base, all.data(), all.size());
        d->d_detail = d->d_base + base.size();
        const std::vector<float> all32(all.begin(), all.end());                                       // round to nearest
        if (e == hipSuccess) e = upload(&d->d_base32, all32.data(), all32.size());
        d->d_detail32 = d->d_base32 + base.size();
    }
    if (e == hipSuccess) e = upload(&d->d_rows, dict.block_rows.data(), dict.block_rows.size());
    if (e == hipSuccess) e = upload(&d->d_rowoff, dict.block_row_off.data(), dict.block_row_off.size());
    // split-bfloat16 filter copies in MFMA operand order (host_dictionary.h: filter_tiles, k order 1): base rows as 32 tiles of
    // 16 rows, every detail block as 4
    std::vector<uint8_t> shadow(3 * det_rows, 0);
    if (e == hipSuccess) {
        const std::vector<uint16_t> base_t = mpc::filter_tiles(dict.base.data(), dict.num_base, mpc::kBaseFilterTiles, 1);
        std::vector<uint16_t> det_t;
        det_t.reserve(3 * static_cast<size_t>(dict.num_base) * mpc::kBlockFilterTiles * mpc::kFilterTileHalves);
        for (int ch = 0; ch < 3; ++ch)
            for (int b = 0; b < dict.num_base; ++b) {
                std::vector<uint8_t> sh;
                const std::vector<uint16_t> t = mpc::filter_tiles(
                    dict.detail[ch].data() + static_cast<size_t>(dict.block_row_off[b]) * mpc::kTileN, dict.block_rows[b],
                    mpc::kBlockFilterTiles, 1, &sh);
                det_t.insert(det_t.end(), t.begin(), t.end());
                std::copy(sh.begin(), sh.end(), shadow.begin() + static_cast<size_t>(ch) * det_rows + static_cast<size_t>(dict.block_row_off[b]));
            }
        e = upload(&d->d_base_t1, base_t.data(), base_t.size());
        if (e == hipSuccess) e = upload(&d->d_detail_t1, det_t.data(), det_t.size());
    }
    if (e == hipSuccess) e = upload(&d->d_shadow, shadow.data(), shadow.size());
    if (e == hipSuccess) e = hipDeviceGetAttribute(&d->num_cus, hipDeviceAttributeMultiprocessorCount, device);
    // Gram table, built on the device
    const long long n_sel = dict.num_base + static_cast<long long>(det_rows), stride = static_cast<long long>(dict.num_base) * 64;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->d_gram), sizeof(float) * 3 * n_sel * stride);
    for (int ch = 0; ch < 3 && e == hipSuccess; ++ch)
        e = static_cast<hipError_t>(mpc::launch_gram(d->d_base, d->d_detail + static_cast<size_t>(ch) * det_rows * mpc::kTileN, d->d_rows,
                                                     d->d_rowoff, d->d_shadow + static_cast<size_t>(ch) * det_rows,
                                                     d->d_gram + static_cast<size_t>(ch) * n_sel * stride, dict.num_base,
                                                     static_cast<int>(n_sel), stride, nullptr));
    // persistent kernel: streams, events, queue words, per-wave scratch for one workgroup per CU
    d->workgroups = d->num_cus > 0 ? d->num_cus : 1;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&d->done[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(d->done[0], nullptr);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->queues), 64);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->stats), 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(d->stats, 0, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->pair_p), sizeof(float) * mpc::pursuit_scratch_floats(d->workgroups));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->pair_meta), sizeof(unsigned) * mpc::pursuit_scratch_meta(d->workgroups));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d->pair_e), sizeof(float) * mpc::pursuit_scratch_bounds(d->workgroups));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return e;
    if (device < 64) g_dicts[device] = d;
    *out = d;
    return hipSuccess;
}
}  // namespace

struct mpc_context {
    int K = 0, block_size = 0, device = -1;
    double bpp = 0.0;
    bool fast = false;                // the `...Fast` (float) flavour of the tile path (mpc_context_set_fast)
    mpc::Dictionary dict;
    std::vector<double> quant;        // [3*K]
    std::shared_ptr<DeviceDict> dd;   // owns the dictionary's device residents; the pointers below alias it
    // device residents (uploaded once)
    double* d_base = nullptr;
    double* d_detail = nullptr;
    double* d_quant = nullptr;        // the context's tables; only mpc_context_set_quant changes them
    // per-call quantiser overrides (the `quant` argument of the encode / decode entry points) go to a ring of device
    // slots of their own, so that a later call with quant == NULL still quantises with the context's tables
    static constexpr int kQuantSlots = 16;
    double* d_quant_ring = nullptr;   // [kQuantSlots][3 * MPC_MAX_K]
    unsigned quant_next = 0;
    std::recursive_mutex host_calls;  // the host-buffer entry points share the staging buffers below
    int32_t* d_rows = nullptr;
    int32_t* d_rowoff = nullptr;
    int* d_flag = nullptr;            // decode: set when a record indexes outside its dictionary
    // grow-only device staging for the host-buffer entry points (mpc_encode_tiles / mpc_encode_image): allocating and
    // freeing five buffers per call cost several times the encode itself
    void* stage = nullptr;
    size_t stage_bytes = 0;
    void* host_stage = nullptr;       // pinned: records of mpc_encode_image on their way to the entropy stage
    size_t host_stage_bytes = 0;
    // mpc_encode_images: upload / compute / download streams and per-slot events (upload done, pursuit done, download done)
    hipStream_t seq_up = nullptr, seq_compute = nullptr;
    bool seq_prioritised = false;                    // the side streams outrank the pursuits' (encode_sequence)
    // read by device-pointer encodes, which do not take `host_calls`: a stale value costs or gains a few workgroups, nothing else
    std::atomic<int> seq_workgroups{0};              // > 0: the pursuits of a frame sequence leave CUs to the kernels behind them
    std::atomic<int> user_workgroups{0};             // > 0: mpc_context_set_tile_encode_workgroups
    static constexpr int kSeqSlots = 6;              // frames in flight in mpc_encode_images (a frame's container is ready about
                                                     // three pursuits after its own started)
    int pipes_cap = 0;                               // > 0: at most this many concurrent sub-batches per call
    hipEvent_t seq_events[kSeqSlots][3] = {};
    static constexpr int kSingleStripes = 4;
    hipEvent_t seq_stripe_up[kSingleStripes] = {};   // a single host frame goes up and is encoded in row stripes (encode_sequence)
    hipEvent_t seq_pursuit_done[kSeqSlots] = {};      // behind a slot's pursuit, for the slot's own stream to wait on
    hipStream_t seq_down[kSeqSlots] = {};             // one download stream per slot: its worker thread drives it
    // device-side entropy stage (mp_entropy.hip): per-slot buffers (grow-only) and the histogram tables all slots share
    // (phase 1 of every frame runs on one stream, in order)
    struct EntropySlot {
        void* dev = nullptr;
        size_t dev_bytes = 0;
        void* host = nullptr;                        // pinned
        size_t host_bytes = 0;
        size_t tiles = 0;                            // geometry the tables inside `dev` were last cleared for
        int K = 0;
    };
    EntropySlot ent[kSeqSlots];
    std::shared_ptr<void> jobs[kSeqSlots];           // ContainerJob (mpc_container_job_*): records on the device -> container, in steps
    // The pursuit of a call is cut into sub-batches that run on `pipes` internal streams, each with its own
    // workspace: the latency-bound bookkeeping kernels of one sub-batch (finish, update, bucket, fill) overlap
    // the machine-filling sweeps of the other.  Fork/join with events on the caller's stream: still no host
    // synchronisation, still capturable.
    struct Pipe {
        hipStream_t stream = nullptr;
        hipStream_t side = nullptr;        // each step's detail branch runs here, beside the base sweep
        hipEvent_t done = nullptr, fork = nullptr, join = nullptr;
        void* mem = nullptr;
        mpc::Workspace ws{};
    };
    std::vector<Pipe> pipes;
    hipEvent_t fork = nullptr;
    int ws_cap = 0;                   // tile-channels per pipe workspace
    int base_rows_padded = 0;
    int max_waves = 0;
    int num_cus = 0;
    // optional live timing of the base-sweep launches (mpc_kernel_timing_*)
    bool timing = false;
    std::vector<hipEvent_t> timing_events;      // 2 per launch, grown on demand
    size_t timing_used = 0;
    hipEvent_t timing_ref = nullptr;            // common time origin for the union of launch intervals
};

namespace {
constexpr long long kMaxBatchDefault = 3LL * 524288;   // tile-channels in flight per call (an 8K frame in one go; ~1.5 GB of workspace)

// tuning overrides for experiments (results never depend on them)
int env_int(const char* name, int fallback) {
    const char* v = std::getenv(name);
    return (v && *v) ? std::atoi(v) : fallback;
}

// MPC_MAX_BATCH_TILES shrinks the in-flight limit (tests use it to exercise the multi-batch path on small frames)
long long max_batch() {
    const int tiles = env_int("MPC_MAX_BATCH_TILES", 0);
    return tiles > 0 ? 3LL * ((tiles + 255) / 256 * 256) : kMaxBatchDefault;
}

// device table a call quantises with: the context's, or a ring slot holding the call's override (copied on `s`)
mpc_status call_quant(mpc_context* c, const double* quant, hipStream_t s, const double** d_q) {
    *d_q = c->d_quant;
    if (!quant) return MPC_OK;
    double* slot = c->d_quant_ring + static_cast<size_t>(c->quant_next++ % mpc_context::kQuantSlots) * 3 * MPC_MAX_K;
    HIP_TRY(hipMemcpyAsync(slot, quant, 3 * sizeof(double) * c->K, hipMemcpyHostToDevice, s));
    *d_q = slot;
    return MPC_OK;
}

mpc::DictDevice dict_device(const mpc_context* c) {
    mpc::DictDevice d{};
    d.base = c->d_base;
    d.num_base = c->dict.num_base;
    d.base_rows_padded = c->base_rows_padded;
    d.detail = c->d_detail;
    d.base32 = c->dd ? c->dd->d_base32 : nullptr;
    d.detail32 = c->dd ? c->dd->d_detail32 : nullptr;
    d.detail_rows = c->dict.total_detail_rows();
    d.block_rows = c->d_rows;
    d.block0_rows = c->dict.block_rows.empty() ? 0 : c->dict.block_rows[0];
    d.block_row_off = c->d_rowoff;
    return d;
}

// grow-only workspaces; allocation synchronises the device, so callers that must not (graph capture)
// call mpc_reserve() first
// how many sub-batches of a call run concurrently (measured on MI355X: 2 for a 1080p frame, 3 from ~300k
// tile-channels up, 4 for an 8K frame)
int pipes_for(const mpc_context* c, long long tile_channels) {
    const int forced = env_int("MPC_PIPES", 0);
    if (forced > 0) return std::min(forced, 4);
    if (tile_channels <= 3 * 4096) return 1;              // do not split what cannot fill the machine
    int pipes = tile_channels >= 1200000 ? 4 : (tile_channels >= 300000 ? 3 : 2);      // 4: an 8K frame
    if (c->pipes_cap > 0 && pipes > c->pipes_cap) pipes = c->pipes_cap;
    return pipes;
}

// the step-synchronous exhaustive sweeps (the product's cross-check of the persistent kernel) run only on request:
// MPC_PATH=steps, or MPC_FILTER=0 (the older name)
bool steps_path() {
    const char* v = std::getenv("MPC_PATH");
    return (v && std::strcmp(v, "steps") == 0) || env_int("MPC_FILTER", 1) == 0;
}

mpc_status ensure_workspace(mpc_context* c, long long tile_channels) {
    if (!steps_path()) return MPC_OK;                 // the persistent kernel's scratch lives in the shared DeviceDict
    const int want_pipes = pipes_for(c, tile_channels);
    const long long kMaxBatch = max_batch();
    long long total = tile_channels < kMaxBatch ? tile_channels : kMaxBatch;
    long long cap = (total + want_pipes - 1) / want_pipes;
    cap = (cap + 767) / 768 * 768;                                        // whole units (3 tile-channels), whole 256-blocks
    if (cap <= c->ws_cap && static_cast<int>(c->pipes.size()) >= want_pipes) return MPC_OK;
    if (hipDeviceSynchronize() != hipSuccess) return fail(MPC_ERR_HIP, "device synchronise failed");
    if (!c->fork && hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess)
        return fail(MPC_ERR_HIP, "event creation failed");
    while (static_cast<int>(c->pipes.size()) < want_pipes) {
        mpc_context::Pipe p;
        if (hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&p.done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&p.join, hipEventDisableTiming) != hipSuccess)
            return fail(MPC_ERR_HIP, "stream/event creation failed");
        c->pipes.push_back(p);
    }
    if (cap < c->ws_cap) cap = c->ws_cap;
    const size_t bytes = mpc::workspace_bytes(static_cast<int>(cap), c->K);
    for (auto& p : c->pipes) {
        if (p.mem && cap == c->ws_cap) continue;          // already large enough
        if (p.mem) (void)hipFree(p.mem);
        p.mem = nullptr;
        hipError_t e = hipMalloc(&p.mem, bytes);
        if (e != hipSuccess) { c->ws_cap = 0; return fail(MPC_ERR_ALLOC, "workspace of %zu bytes: %s", bytes, hipGetErrorString(e)); }
        p.ws = mpc::carve_workspace(p.mem, static_cast<int>(cap), c->K);
    }
    c->ws_cap = static_cast<int>(cap);
    return MPC_OK;
}

// Sweep work is cut fine (8 atom ranges per 64 tile-channels, 4 row ranges per detail block) and handed to
// machine-sized persistent grids, so a step's last round is nearly full whatever the active count is.
constexpr int kBaseParts = 8;
constexpr int kRowParts = 4;


// The persistent path (mp_pursuit.hip): ONE launch on the caller's stream runs all K steps of every tile-channel; its
// workgroups are split over the channels (a workgroup's LDS holds one channel's DetailBasis[0]).  No host synchronisation,
// no allocation: graph-capturable.  Launches of one process are serialised on the device by a lock-ordered event chain,
// because they share the per-device scratch and queue words.
mpc_status run_persistent(mpc_context* c, const mpc::FrameInput& in, const mpc::Outputs& out, const double* d_quant,
                          long long total_tc, void* stream) {
    DeviceDict& d = *c->dd;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = in.vec_in != nullptr;
    const long long n_tc = vec ? total_tc : total_tc / 3;
    const long long n_units = (n_tc + 15) / 16 * (vec ? 1 : 3);         // groups of 16 tile-channels, all channels
    if (n_tc >= (1LL << 31)) return fail(MPC_ERR_ARGUMENT, "batch too large");
    // One workgroup fits a CU (its LDS holds the dictionary); its waves start on luma and move on to the chroma channels as the
    // queues run dry (mp_pursuit.hip: channel switch), so a small frame spreads over the channels by itself.
    const int per_wg = mpc::pursuit_units_per_workgroup();
    int workgroups = static_cast<int>(std::min<long long>((n_units + per_wg - 1) / per_wg, d.workgroups));
    if (const int limit = c->seq_workgroups.load(std::memory_order_relaxed); limit > 0) workgroups = std::min(workgroups, limit);
    if (const int limit = c->user_workgroups.load(std::memory_order_relaxed); limit > 0) workgroups = std::min(workgroups, limit);
    const int forced = env_int("MPC_WORKGROUPS", 0);
    if (forced > 0) workgroups = std::min(forced, d.workgroups);
    std::lock_guard<std::mutex> hold(d.launch_lock);
    HIP_TRY(hipStreamWaitEvent(s, d.done[0], 0));             // the previous launch of this process (any stream) has drained
    HIP_TRY(hipMemsetAsync(d.queues, 0, 3 * sizeof(unsigned), s));
    const long long n_sel = d.num_base + d.detail_rows, stride = static_cast<long long>(d.num_base) * 64;
    mpc::PursuitArgs a{};
    a.base = d.d_base;
    a.base32 = d.d_base32;

    a.fast = c->fast ? 1 : 0;
    a.base_tiles = d.d_base_t1;
    for (int ch = 0; ch < 3; ++ch) {
        a.detail[ch] = d.d_detail + static_cast<size_t>(ch) * d.detail_rows * mpc::kTileN;
        a.detail32[ch] = d.d_detail32 + static_cast<size_t>(ch) * d.detail_rows * mpc::kTileN;
        a.block_tiles[ch] = d.d_detail_t1 + static_cast<size_t>(ch) * d.num_base * mpc::kBlockFilterTiles * mpc::kFilterTileHalves;
        a.gram[ch] = d.d_gram + static_cast<size_t>(ch) * n_sel * stride;
        a.n_tc[ch] = vec ? (ch == in.vec_channel ? n_tc : 0) : n_tc;
    }
    a.pair_p = d.pair_p;
    a.pair_meta = d.pair_meta;
    a.pair_e = d.pair_e;
    a.workgroups = workgroups;
    a.gram_stride = stride;
    a.block_rows = d.d_rows;
    a.block_row_off = d.d_rowoff;
    a.quant = d_quant;
    a.K = c->K;
    a.num_base = d.num_base;
    a.rows0 = c->dict.block_rows.empty() ? 0 : c->dict.block_rows[0];
    a.rgb = in.rgb;
    a.width = in.width;
    a.height = in.height;
    a.row_stride = in.row_stride;
    a.frame_stride = in.frame_stride;
    a.tile_row_begin = in.tile_row_begin;
    a.tile_rows = in.tile_rows;
    a.tiles_x = in.tiles_x;
    a.out_tile_rows = in.out_tile_rows;
    a.rgb_aligned8 = (reinterpret_cast<uintptr_t>(in.rgb) % 8 == 0 && in.row_stride % 8 == 0 && (in.frames <= 1 || in.frame_stride % 8 == 0)) ? 1 : 0;
    a.vec_in = in.vec_in;
    a.vec_channel = in.vec_channel;
    a.queue = d.queues;
    a.out = out;
    a.stats = d.stats;
    hipEvent_t* ev = nullptr;
    if (c->timing) {
        const size_t need = c->timing_used + 2;
        while (c->timing_events.size() < need) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return fail(MPC_ERR_HIP, "hipEventCreate failed");
            c->timing_events.push_back(e);
        }
        ev = c->timing_events.data() + c->timing_used;
        c->timing_used = need;
        HIP_TRY(hipEventRecord(ev[0], s));
    }
#ifdef MPC_STAMPS
    constexpr int kDebugWords = 24 + 2 * 1024;
    static unsigned long long* d_debug = nullptr;
    if (!d_debug) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_debug), kDebugWords * sizeof(unsigned long long)));
    {
        std::vector<unsigned long long> init(kDebugWords, 0ULL);
        for (int b = 0; b < 1024; ++b) init[24 + 2 * b] = ~0ULL;
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(d_debug, init.data(), kDebugWords * sizeof(unsigned long long), hipMemcpyHostToDevice));
    }
    a.debug = d_debug;
#endif
    const int err = mpc::launch_pursuit(a, s);
    if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    if (ev) HIP_TRY(hipEventRecord(ev[1], s));
    HIP_TRY(hipEventRecord(d.done[0], s));
#ifdef MPC_STAMPS
    {
        std::vector<unsigned long long> all(kDebugWords);
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(all.data(), d_debug, kDebugWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        const unsigned long long* hst = all.data();
        unsigned long long tot = 0;
        for (int i = 0; i < 12; ++i) tot += hst[i];
        std::fprintf(stderr, "[stamps wg%d] wave-steps %llu live-lanes/step %.1f pass2-groups %llu rounds %llu exhaustive %llu | cycles/wave-step:",
                     workgroups, hst[12], hst[12] ? (double)hst[16] / hst[12] / 4.0 : 0.0, hst[13], hst[14], hst[15]);
        for (int i = 0; i < 12; ++i) std::fprintf(stderr, " %d:%.0f", i, hst[12] ? (double)hst[i] / hst[12] : 0.0);
        std::fprintf(stderr, " total %.0f | pair rounds/step %.2f slots/round %.2f new pairs/step %.2f\n", hst[12] ? (double)tot / hst[12] : 0.0,
                     hst[12] ? (double)hst[17] / hst[12] : 0.0, hst[17] ? (double)hst[18] / hst[17] : 0.0, hst[12] ? (double)hst[19] / hst[12] : 0.0);
        // workgroup residency: how many workgroups are on the machine over the launch (20 slices), and per channel when its
        // first / last workgroup came and went (microseconds from the first workgroup's start; 100 MHz counter)
        const int grid = a.workgroups;
        unsigned long long t0 = ~0ULL, t1 = 0;
        for (int b = 0; b < grid && b < 1024; ++b) { t0 = std::min(t0, all[24 + 2 * b]); t1 = std::max(t1, all[25 + 2 * b]); }
        if (t1 > t0) {
            const double span = (double)(t1 - t0);
            double busy = 0.0;
            int slices[20] = {};
            for (int b = 0; b < grid && b < 1024; ++b) {
                busy += (double)(all[25 + 2 * b] - all[24 + 2 * b]);
                for (int k = 0; k < 20; ++k) {
                    const double mid = t0 + span * (k + 0.5) / 20.0;
                    if ((double)all[24 + 2 * b] <= mid && mid < (double)all[25 + 2 * b]) ++slices[k];
                }
            }
            std::fprintf(stderr, "[residency] span %.1f us, workgroup-time / (%d CUs x span) = %.3f | resident workgroups per 5 %% slice:", span / 100.0,
                         d.workgroups, busy / (span * d.workgroups));
            for (int k = 0; k < 20; ++k) std::fprintf(stderr, " %d", slices[k]);
            std::fprintf(stderr, "\n");
        }
    }
#endif
    return MPC_OK;
}

mpc_status run_pursuit(mpc_context* c, const mpc::FrameInput& in, const mpc::Outputs& out, const double* d_quant,
                       long long total_tc, void* stream) {
    // MPC_PATH=steps / MPC_FILTER=0: the step-synchronous exhaustive double sweeps of mp_kernels.hip (the product's own
    // cross-check); default: the persistent kernel
    if (!steps_path()) return run_persistent(c, in, out, d_quant, total_tc, stream);
    if (c->fast) return fail(MPC_ERR_ARGUMENT, "the float flavour runs on the persistent kernel only (unset MPC_PATH / MPC_FILTER)");
    mpc_status st = ensure_workspace(c, total_tc);
    if (st != MPC_OK) return st;
    const mpc::DictDevice dict = dict_device(c);
    hipStream_t caller = static_cast<hipStream_t>(stream);
    // sub-batch size: an even share per pipe (whole units), at most the workspace capacity
    const long long npipes = pipes_for(c, total_tc);
    long long share = (total_tc + npipes - 1) / npipes;
    share = (share + 767) / 768 * 768;
    if (share > c->ws_cap) share = c->ws_cap;
    HIP_TRY(hipEventRecord(c->fork, caller));
    size_t used_pipes = 0;
    long long index = 0;
    for (long long begin = 0; begin < total_tc; begin += share, ++index) {
        const long long n = (total_tc - begin < share) ? total_tc - begin : share;
        auto& pipe = c->pipes[static_cast<size_t>(index % npipes)];
        if (index < npipes) {
            HIP_TRY(hipStreamWaitEvent(pipe.stream, c->fork, 0));
            ++used_pipes;
        }
        void** events = nullptr;
        if (c->timing) {
            const size_t need = c->timing_used + 2 * static_cast<size_t>(c->K);
            while (c->timing_events.size() < need) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return fail(MPC_ERR_HIP, "hipEventCreate failed");
                c->timing_events.push_back(e);
            }
            events = reinterpret_cast<void**>(c->timing_events.data() + c->timing_used);
            c->timing_used = need;
        }
        const int err = mpc::enqueue_pursuit(dict, pipe.ws, in, out, d_quant, c->K, begin, static_cast<int>(n),
                                             env_int("MPC_BASE_PARTS", kBaseParts), env_int("MPC_ROW_PARTS", kRowParts),
                                             env_int("MPC_SWEEP_WAVES", c->max_waves), pipe.stream, events,
                                             env_int("MPC_SIDE", 0) ? pipe.side : nullptr, pipe.fork, pipe.join);
        if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    }
    for (size_t i = 0; i < used_pipes; ++i) {
        HIP_TRY(hipEventRecord(c->pipes[i].done, c->pipes[i].stream));
        HIP_TRY(hipStreamWaitEvent(caller, c->pipes[i].done, 0));
    }
    return MPC_OK;
}
}  // namespace

extern "C" {

const char* mpc_version(void) { return "mpcodec 0.1 (gfx950)"; }
const char* mpc_last_error(void) { return g_error; }
void mpc_set_error_text(const char* text) { std::snprintf(g_error, sizeof g_error, "%s", text ? text : ""); }

mpc_status mpc_context_create(int K, int block_size, double bpp, int device, mpc_context** out) {
    return guarded([&]() -> mpc_status {
    if (!out) return fail(MPC_ERR_ARGUMENT, "out is null");
    *out = nullptr;
    if (K < 1 || K > MPC_MAX_K) return fail(MPC_ERR_ARGUMENT, "K=%d out of range 1..%d", K, MPC_MAX_K);
    if (block_size < 1 || block_size > 8) return fail(MPC_ERR_ARGUMENT, "block size %d out of range 1..8", block_size);
    if (device >= 0 && block_size != 8)
        return fail(MPC_ERR_ARGUMENT, "the device path implements 8x8 tiles only (got %d)", block_size);
    mpc_context* c = new (std::nothrow) mpc_context;
    if (!c) return fail(MPC_ERR_ALLOC, "out of memory");
    try {
        c->K = K;
        c->block_size = block_size;
        c->bpp = bpp;
        c->dict = mpc::build_dictionary(block_size);
        c->quant.resize(3 * static_cast<size_t>(K));
        mpc::quantisation_tables(K, block_size, bpp, c->quant.data());
    } catch (const std::exception& e) {
        delete c;
        return fail(MPC_ERR_ARGUMENT, "%s", e.what());
    }
    c->device = device;
    if (device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) {
            delete c;
            return fail(MPC_ERR_NO_DEVICE, "HIP device %d not available (%d devices visible)", device, ndev);
        }
        hipError_t e = acquire_device_dict(device, c->dict, &c->dd);
        if (e == hipSuccess) {
            c->d_base = c->dd->d_base;
            c->d_detail = c->dd->d_detail;
            c->d_rows = c->dd->d_rows;
            c->d_rowoff = c->dd->d_rowoff;
            c->base_rows_padded = c->dd->base_rows_padded;
        }
        if (e == hipSuccess) e = upload(&c->d_quant, c->quant.data(), c->quant.size());
        if (e == hipSuccess)
            e = hipMalloc(reinterpret_cast<void**>(&c->d_quant_ring), sizeof(double) * mpc_context::kQuantSlots * 3 * MPC_MAX_K);
        if (e != hipSuccess) {
            mpc_context_destroy(c);
            return fail(MPC_ERR_HIP, "device setup failed: %s", hipGetErrorString(e));
        }
        (void)hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device);
        c->max_waves = 12 * c->num_cus;
    }
    *out = c;
    return MPC_OK;
    });
}

void mpc_context_destroy(mpc_context* c) {
    if (!c) return;
    if (c->device >= 0) {
        (void)hipSetDevice(c->device);
        (void)hipFree(c->d_quant);
        (void)hipFree(c->d_quant_ring);
        (void)hipFree(c->d_flag);
        (void)hipFree(c->stage);
        if (c->host_stage) (void)hipHostFree(c->host_stage);
        for (auto& e : c->ent) {
            if (e.dev) (void)hipFree(e.dev);
            if (e.host) (void)hipHostFree(e.host);
        }
        if (c->seq_up) (void)hipStreamDestroy(c->seq_up);
        if (c->seq_compute) (void)hipStreamDestroy(c->seq_compute);
        for (hipStream_t sd : c->seq_down)
            if (sd) (void)hipStreamDestroy(sd);
        for (auto& slot : c->seq_events)
            for (hipEvent_t e : slot)
                if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->seq_pursuit_done)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : c->seq_stripe_up)
            if (e) (void)hipEventDestroy(e);
        for (auto& p : c->pipes) {
            if (p.mem) (void)hipFree(p.mem);
            if (p.stream) (void)hipStreamDestroy(p.stream);
            if (p.side) (void)hipStreamDestroy(p.side);
            if (p.done) (void)hipEventDestroy(p.done);
            if (p.fork) (void)hipEventDestroy(p.fork);
            if (p.join) (void)hipEventDestroy(p.join);
        }
        if (c->fork) (void)hipEventDestroy(c->fork);
        for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
        if (c->timing_ref) (void)hipEventDestroy(c->timing_ref);
    }
    delete c;
}

int mpc_context_K(const mpc_context* c) { return c ? c->K : 0; }
int mpc_context_block_size(const mpc_context* c) { return c ? c->block_size : 0; }
int mpc_context_num_base(const mpc_context* c) { return c ? c->dict.num_base : 0; }
int mpc_context_detail_rows(const mpc_context* c) { return c ? c->dict.total_detail_rows() : 0; }
int mpc_context_device(const mpc_context* c) { return c ? c->device : -1; }
int mpc_context_max_waves(const mpc_context* c) { return c ? c->max_waves : 0; }

mpc_status mpc_context_set_tile_encode_workgroups(mpc_context* c, int workgroups) {
    if (!c || workgroups < 0) return fail(MPC_ERR_ARGUMENT, "bad argument");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    c->user_workgroups = workgroups;
    return MPC_OK;
}

mpc_status mpc_context_set_fast(mpc_context* c, int on) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    c->fast = on != 0;
    return MPC_OK;
}
int mpc_context_is_fast(const mpc_context* c) { return c && c->fast ? 1 : 0; }

mpc_status mpc_context_get_quant(const mpc_context* c, double* quant) {
    if (!c || !quant) return fail(MPC_ERR_ARGUMENT, "null argument");
    std::memcpy(quant, c->quant.data(), c->quant.size() * sizeof(double));
    return MPC_OK;
}

mpc_status mpc_context_set_quant(mpc_context* c, const double* quant) {
    if (!c || !quant) return fail(MPC_ERR_ARGUMENT, "null argument");
    std::memcpy(c->quant.data(), quant, c->quant.size() * sizeof(double));
    if (c->device >= 0) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemcpy(c->d_quant, quant, c->quant.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return MPC_OK;
}

mpc_status mpc_context_get_dictionary(const mpc_context* c, double* base, int32_t* block_rows, double* dy, double* du,
                                      double* dv) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    if (base) std::memcpy(base, c->dict.base.data(), c->dict.base.size() * sizeof(double));
    if (block_rows) std::memcpy(block_rows, c->dict.block_rows.data(), c->dict.block_rows.size() * sizeof(int32_t));
    double* det[3] = {dy, du, dv};
    for (int ch = 0; ch < 3; ++ch)
        if (det[ch]) std::memcpy(det[ch], c->dict.detail[ch].data(), c->dict.detail[ch].size() * sizeof(double));
    return MPC_OK;
}

// whole_frame_order: the records go where one launch over the whole frame would put them (FrameInput::out_tile_rows) and the
// caller has zeroed d_choices for the whole frame (stripes of one frame encoded one by one, encode_sequence's single frames)
static mpc_status encode_batch_device(mpc_context* c, const uint8_t* d_rgb, int frames, size_t frame_stride, int width,
                                      int height, size_t row_stride, int tile_row_begin, int tile_row_end,
                                      const double* quant, uint16_t* d_counts, mpc_basis_choice* d_choices,
                                      double* d_energy, uint32_t* d_swept, void* stream, bool whole_frame_order);

mpc_status mpc_encode_batch_device(mpc_context* c, const uint8_t* d_rgb, int frames, size_t frame_stride, int width,
                                   int height, size_t row_stride, int tile_row_begin, int tile_row_end,
                                   const double* quant, uint16_t* d_counts, mpc_basis_choice* d_choices,
                                   double* d_energy, uint32_t* d_swept, int waves, void* stream) {
    (void)waves;
    return encode_batch_device(c, d_rgb, frames, frame_stride, width, height, row_stride, tile_row_begin, tile_row_end, quant, d_counts,
                               d_choices, d_energy, d_swept, stream, false);
}

static mpc_status encode_batch_device(mpc_context* c, const uint8_t* d_rgb, int frames, size_t frame_stride, int width,
                                      int height, size_t row_stride, int tile_row_begin, int tile_row_end,
                                      const double* quant, uint16_t* d_counts, mpc_basis_choice* d_choices,
                                      double* d_energy, uint32_t* d_swept, void* stream, bool whole_frame_order) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    if (!d_rgb || !d_counts || !d_choices) return fail(MPC_ERR_ARGUMENT, "null buffer");
    if (width < 1 || height < 1 || row_stride < static_cast<size_t>(3) * width)
        return fail(MPC_ERR_ARGUMENT, "bad geometry %dx%d stride %zu", width, height, row_stride);
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    if (tile_row_begin < 0 || tile_row_end > tiles_y || tile_row_begin >= tile_row_end)
        return fail(MPC_ERR_ARGUMENT, "tile rows [%d,%d) outside 0..%d", tile_row_begin, tile_row_end, tiles_y);
    if (frames < 1 || (frames > 1 && frame_stride < row_stride * static_cast<size_t>(height)))
        return fail(MPC_ERR_ARGUMENT, "bad batch: %d frames, stride %zu", frames, frame_stride);
    const long long tiles = static_cast<long long>(tiles_x) * (tile_row_end - tile_row_begin) * frames;
    if (tiles * 3 >= (1LL << 31)) return fail(MPC_ERR_ARGUMENT, "batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(c->device));
    const double* d_q = nullptr;
    if (const mpc_status qs = call_quant(c, quant, s, &d_q); qs != MPC_OK) return qs;
    if (!whole_frame_order) HIP_TRY(hipMemsetAsync(d_choices, 0, sizeof(mpc_basis_choice) * tiles * 3 * c->K, s));
    mpc::FrameInput in{};
    in.rgb = d_rgb;
    in.width = width;
    in.height = height;
    in.row_stride = static_cast<long long>(row_stride);
    in.frames = frames;
    in.frame_stride = static_cast<long long>(frame_stride);
    in.tile_row_begin = tile_row_begin;
    in.tile_rows = tile_row_end - tile_row_begin;
    in.tiles_x = tiles_x;
    in.out_tile_rows = whole_frame_order ? tiles_y : 0;
    in.vec_in = nullptr;
    in.vec_channel = 0;
    mpc::Outputs out{};
    out.counts = d_counts;
    out.choices = reinterpret_cast<uint32_t*>(d_choices);
    out.energy = d_energy;
    out.swept = d_swept;
    return run_pursuit(c, in, out, d_q, tiles * 3, stream);
}

mpc_status mpc_encode_tiles_device(mpc_context* c, const uint8_t* d_rgb, int width, int height, size_t row_stride,
                                   int tile_row_begin, int tile_row_end, const double* quant, uint16_t* d_counts,
                                   mpc_basis_choice* d_choices, double* d_energy, uint32_t* d_swept, int waves,
                                   void* stream) {
    return mpc_encode_batch_device(c, d_rgb, 1, 0, width, height, row_stride, tile_row_begin, tile_row_end, quant,
                                   d_counts, d_choices, d_energy, d_swept, waves, stream);
}

// upload, pursuit, download through the context's staging area; planar = the records come back as [3][K][tiles]
static mpc_status encode_tiles_staged(mpc_context* c, const uint8_t* rgb, int width, int height, size_t row_stride,
                                      int tile_row_begin, int tile_row_end, const double* quant, uint16_t* counts,
                                      mpc_basis_choice* choices, double* energy, uint32_t* swept, bool planar) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    if (!rgb || !counts || !choices) return fail(MPC_ERR_ARGUMENT, "null buffer");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    HIP_TRY(hipSetDevice(c->device));
    const int tiles_x = (width + 7) / 8;
    const long long tiles = static_cast<long long>(tiles_x) * (tile_row_end - tile_row_begin);
    if (tiles <= 0) return fail(MPC_ERR_ARGUMENT, "empty stripe");
    const size_t img_bytes = row_stride * static_cast<size_t>(height);
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t n_tc = static_cast<size_t>(tiles) * 3;
    const size_t off_counts = up(img_bytes), off_choices = off_counts + up(sizeof(uint16_t) * n_tc),
                 off_energy = off_choices + up(sizeof(mpc_basis_choice) * n_tc * c->K), off_swept = off_energy + up(sizeof(double) * n_tc),
                 off_planar = off_swept + up(sizeof(uint32_t) * n_tc),
                 total = off_planar + (planar ? up(sizeof(mpc_basis_choice) * n_tc * c->K) : 0);
    if (total > c->stage_bytes) {
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_bytes = 0;
        const hipError_t ea = hipMalloc(&c->stage, total);
        if (ea != hipSuccess) return fail(MPC_ERR_ALLOC, "staging of %zu bytes: %s", total, hipGetErrorString(ea));
        c->stage_bytes = total;
    }
    char* base = static_cast<char*>(c->stage);
    uint8_t* d_rgb = reinterpret_cast<uint8_t*>(base);
    uint16_t* d_counts = reinterpret_cast<uint16_t*>(base + off_counts);
    mpc_basis_choice* d_choices = reinterpret_cast<mpc_basis_choice*>(base + off_choices);
    double* d_energy = reinterpret_cast<double*>(base + off_energy);
    uint32_t* d_swept = reinterpret_cast<uint32_t*>(base + off_swept);
    mpc_status st = MPC_OK;
    hipError_t e = hipMemcpy(d_rgb, rgb, img_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        st = mpc_encode_tiles_device(c, d_rgb, width, height, row_stride, tile_row_begin, tile_row_end, quant, d_counts,
                                     d_choices, d_energy, d_swept, 0, nullptr);
        const mpc_basis_choice* d_records = d_choices;
        if (st == MPC_OK && planar) {
            uint32_t* d_planar = reinterpret_cast<uint32_t*>(base + off_planar);
            const int err = mpc::launch_planar_records(reinterpret_cast<const uint32_t*>(d_choices), d_planar, tiles, c->K, nullptr);
            if (err != 0) st = fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
            d_records = reinterpret_cast<const mpc_basis_choice*>(d_planar);
        }
        if (st == MPC_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(counts, d_counts, sizeof(uint16_t) * n_tc, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(choices, d_records, sizeof(mpc_basis_choice) * n_tc * c->K, hipMemcpyDeviceToHost);
            if (e == hipSuccess && energy) e = hipMemcpy(energy, d_energy, sizeof(double) * n_tc, hipMemcpyDeviceToHost);
            if (e == hipSuccess && swept) e = hipMemcpy(swept, d_swept, sizeof(uint32_t) * n_tc, hipMemcpyDeviceToHost);
        }
    }
    if (st != MPC_OK) return st;
    if (e != hipSuccess) return fail(MPC_ERR_HIP, "HIP failure: %s", hipGetErrorString(e));
    return MPC_OK;
}

mpc_status mpc_encode_tiles(mpc_context* c, const uint8_t* rgb, int width, int height, size_t row_stride,
                            int tile_row_begin, int tile_row_end, const double* quant, uint16_t* counts,
                            mpc_basis_choice* choices, double* energy, uint32_t* swept) {
    return encode_tiles_staged(c, rgb, width, height, row_stride, tile_row_begin, tile_row_end, quant, counts, choices, energy, swept,
                               false);
}

mpc_status mpc_histogram_device(mpc_context* c, const uint16_t* d_counts, const mpc_basis_choice* d_choices,
                                long long tiles, uint32_t* d_hist, void* stream) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    if (!d_counts || !d_choices || !d_hist || tiles < 1) return fail(MPC_ERR_ARGUMENT, "bad argument");
    mpc::HistParams h{};
    h.counts = d_counts;
    h.choices = reinterpret_cast<const uint32_t*>(d_choices);
    h.tiles = tiles;
    h.K = c->K;
    h.hist = d_hist;
    const int err = mpc::launch_histogram(h, stream);
    if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    return MPC_OK;
}

mpc_status mpc_calc_mp_batch(mpc_context* c, int channel, const double* quant_k, const double* inputs, int count,
                             mpc_basis_choice* choices, uint16_t* counts, double* energy, uint32_t* swept) {
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    if (channel < 0 || channel > 2 || !inputs || !choices || !counts || count < 1)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const int K = c->K;
    double* d_in = nullptr;
    uint16_t* d_counts = nullptr;
    uint32_t* d_choices = nullptr;
    double* d_energy = nullptr;
    uint32_t* d_swept = nullptr;
    double* d_q = nullptr;
    std::vector<double> q(c->quant);
    if (quant_k) std::memcpy(q.data() + static_cast<size_t>(channel) * K, quant_k, sizeof(double) * K);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_in), sizeof(double) * 64 * count);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_counts), sizeof(uint16_t) * count);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_choices), sizeof(uint32_t) * count * K);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_energy), sizeof(double) * count);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_swept), sizeof(uint32_t) * count);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_q), sizeof(double) * 3 * K);
    if (e == hipSuccess) e = hipMemcpy(d_in, inputs, sizeof(double) * 64 * count, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q, q.data(), sizeof(double) * 3 * K, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_choices, 0, sizeof(uint32_t) * count * K);
    mpc_status st = MPC_OK;
    if (e == hipSuccess) {
        mpc::FrameInput in{};
        in.vec_in = d_in;
        in.vec_channel = channel;
        in.frames = 1;
        mpc::Outputs out{};
        out.counts = d_counts;
        out.choices = d_choices;
        out.energy = d_energy;
        out.swept = d_swept;
        st = run_pursuit(c, in, out, d_q, count, nullptr);
        if (st == MPC_OK) e = hipDeviceSynchronize();
        if (st == MPC_OK && e == hipSuccess) e = hipMemcpy(counts, d_counts, sizeof(uint16_t) * count, hipMemcpyDeviceToHost);
        if (st == MPC_OK && e == hipSuccess)
            e = hipMemcpy(choices, d_choices, sizeof(uint32_t) * count * K, hipMemcpyDeviceToHost);
        if (st == MPC_OK && e == hipSuccess && energy) e = hipMemcpy(energy, d_energy, sizeof(double) * count, hipMemcpyDeviceToHost);
        if (st == MPC_OK && e == hipSuccess && swept) e = hipMemcpy(swept, d_swept, sizeof(uint32_t) * count, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_in);
    (void)hipFree(d_counts);
    (void)hipFree(d_choices);
    (void)hipFree(d_energy);
    (void)hipFree(d_swept);
    (void)hipFree(d_q);
    if (st != MPC_OK) return st;
    if (e != hipSuccess) return fail(MPC_ERR_HIP, "HIP failure: %s", hipGetErrorString(e));
    return MPC_OK;
}

// live timing of the dominant kernel (mp_pursuit_kernel; mp_base_kernel with MPC_PATH=steps) with HIP events on the launch stream
void mpc_kernel_timing_enable(mpc_context* c, int on) {
    if (!c) return;
    c->timing = on != 0;
    c->timing_used = 0;
    if (c->timing && c->device >= 0) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        if (c->dd) (void)hipMemset(c->dd->stats, 0, 2 * sizeof(unsigned long long));
        if (!c->timing_ref) (void)hipEventCreate(&c->timing_ref);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(c->timing_ref, nullptr);
        (void)hipEventSynchronize(c->timing_ref);
    }
}

mpc_status mpc_kernel_timing_read(mpc_context* c, double* total_ms, long long* launches, double* busy_ms) {
    if (!c || !total_ms || !launches) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    double sum = 0.0;
    std::vector<std::pair<float, float>> spans;
    for (size_t i = 0; i + 1 < c->timing_used; i += 2) {
        float a = 0.f, b = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, c->timing_ref, c->timing_events[i]));
        HIP_TRY(hipEventElapsedTime(&b, c->timing_ref, c->timing_events[i + 1]));
        sum += b - a;
        spans.emplace_back(a, b);
    }
    // launches on the internal streams overlap: the union of their intervals is the time the machine spent in
    // this kernel
    std::sort(spans.begin(), spans.end());
    double busy = 0.0;
    float lo = 0.f, hi = -1.f;
    for (const auto& sp : spans) {
        if (hi < lo || sp.first > hi) {
            if (hi >= lo) busy += hi - lo;
            lo = sp.first;
            hi = sp.second;
        } else if (sp.second > hi) {
            hi = sp.second;
        }
    }
    if (hi >= lo) busy += hi - lo;
    *total_ms = sum;
    *launches = static_cast<long long>(c->timing_used / 2);
    if (busy_ms) *busy_ms = busy;
    c->timing_used = 0;
    return MPC_OK;
}

mpc_status mpc_kernel_counters_read(mpc_context* c, unsigned long long* mfma_instructions, unsigned long long* tile_channel_steps) {
    if (!c || !mfma_instructions || !tile_channel_steps) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipMemcpy(v, c->dd->stats, sizeof v, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(c->dd->stats, 0, sizeof v));
    *mfma_instructions = v[0];
    *tile_channel_steps = v[1];
    return MPC_OK;
}

mpc_status mpc_reserve(mpc_context* c, long long max_tiles) {
    if (!c || max_tiles < 1) return fail(MPC_ERR_ARGUMENT, "bad argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    HIP_TRY(hipSetDevice(c->device));
    return ensure_workspace(c, max_tiles * 3);
}

mpc_status mpc_calc_mp(mpc_context* c, int channel, const double* quant_k, const double* input64,
                       mpc_basis_choice* choices, int* count) {
    if (!count) return fail(MPC_ERR_ARGUMENT, "null count");
    uint16_t n = 0;
    mpc_status st = mpc_calc_mp_batch(c, channel, quant_k, input64, 1, choices, &n, nullptr, nullptr);
    if (st == MPC_OK) *count = n;
    return st;
}

// ------------------------------------------------------------------------------------------------
// host entropy stage / container
// ------------------------------------------------------------------------------------------------
struct mpc_streams {
    mpc::Streams s;
};

namespace {
uint8_t* give_bytes(const std::vector<uint8_t>& v, size_t* n) {
    uint8_t* p = static_cast<uint8_t*>(std::malloc(v.empty() ? 1 : v.size()));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size());
    *n = v.size();
    return p;
}
uint16_t* give_u16(const std::vector<uint16_t>& v, size_t* n) {
    uint16_t* p = static_cast<uint16_t*>(std::malloc(v.empty() ? 2 : v.size() * 2));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * 2);
    *n = v.size();
    return p;
}
}  // namespace

void mpc_free(void* p) { std::free(p); }

mpc_status mpc_write_compressed(int width, int height, int K, int block_size, const double* quant,
                                const uint16_t* lengths, size_t n_lengths, const uint16_t* const* codes,
                                const size_t* code_lengths, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!quant || !bytes || !nbytes || (!lengths && n_lengths) || !codes || !code_lengths || K < 1 || K > MPC_MAX_K)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    mpc::Streams s;
    s.width = width; s.height = height; s.K = K; s.block_size = block_size;
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < K; ++i) s.quant[ch][i] = static_cast<uint16_t>(quant[ch * K + i]);
    s.lengths.assign(lengths, lengths + n_lengths);
    s.codes.resize(static_cast<size_t>(6 * K));
    for (int i = 0; i < 6 * K; ++i) s.codes[i].assign(codes[i], codes[i] + code_lengths[i]);
    *bytes = give_bytes(mpc::write_compressed(s), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_assemble_streams(int width, int height, int K, int block_size, const double* quant,
                                const uint16_t* counts, const mpc_basis_choice* choices, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!quant || !counts || !choices || !bytes || !nbytes || K < 1 || K > MPC_MAX_K || block_size < 1 || width < 1 || height < 1)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    *bytes = mpc::encode_records_malloc(width, height, K, block_size, quant, counts, reinterpret_cast<const uint32_t*>(choices), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_assemble_planar_streams(int width, int height, int K, int block_size, const double* quant,
                                       const uint16_t* counts, const mpc_basis_choice* planar, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!quant || !counts || !planar || !bytes || !nbytes || K < 1 || K > MPC_MAX_K || block_size < 1 || width < 1 || height < 1)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    *bytes = mpc::encode_planar_records_malloc(width, height, K, block_size, quant, counts, reinterpret_cast<const uint32_t*>(planar), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_assemble_symbol_streams(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                       const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!quant || !counts || (!symbols && stream_off && stream_off[6 * K]) || !stream_off || !bytes || !nbytes || K < 1 || K > MPC_MAX_K ||
        block_size < 1 || width < 1 || height < 1)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    for (int s = 0; s < 6 * K; ++s)
        if (stream_off[s + 1] < stream_off[s]) return fail(MPC_ERR_ARGUMENT, "stream offsets must not decrease");
    *bytes = mpc::encode_symbol_streams_malloc(width, height, K, block_size, quant, counts, symbols, stream_off, nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_assemble_symbol_streams_by_plan(int width, int height, int K, int block_size, const double* quant, const uint16_t* counts,
                                               const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!quant || !counts || (!symbols && stream_off && stream_off[6 * K]) || !stream_off || !bytes || !nbytes || K < 1 || K > MPC_MAX_K ||
        block_size < 1 || width < 1 || height < 1)
        return fail(MPC_ERR_ARGUMENT, "bad argument");
    for (int s = 0; s < 6 * K; ++s)
        if (stream_off[s + 1] < stream_off[s]) return fail(MPC_ERR_ARGUMENT, "stream offsets must not decrease");
    *bytes = mpc::encode_symbol_streams_by_plan_malloc(width, height, K, block_size, quant, counts, symbols, stream_off, nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory or inconsistent plan");
    });
}

mpc_status mpc_read_compressed(const uint8_t* bytes, size_t nbytes, mpc_streams** out) {
    return guarded([&]() -> mpc_status {
    if (!bytes || !out) return fail(MPC_ERR_ARGUMENT, "null argument");
    std::unique_ptr<mpc_streams> h(new mpc_streams);
    if (!mpc::read_compressed(bytes, nbytes, h->s)) return fail(MPC_ERR_BITSTREAM, "Invalid input data");
    *out = h.release();
    return MPC_OK;
    });
}

mpc_status mpc_streams_info(const mpc_streams* h, int* width, int* height, int* K, int* block_size) {
    if (!h) return fail(MPC_ERR_ARGUMENT, "null streams");
    if (width) *width = h->s.width;
    if (height) *height = h->s.height;
    if (K) *K = h->s.K;
    if (block_size) *block_size = h->s.block_size;
    return MPC_OK;
}

mpc_status mpc_streams_quant(const mpc_streams* h, uint16_t* quant) {
    if (!h || !quant) return fail(MPC_ERR_ARGUMENT, "null argument");
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < h->s.K; ++i) quant[ch * h->s.K + i] = h->s.quant[ch][i];
    return MPC_OK;
}

size_t mpc_streams_length(const mpc_streams* h, int index) {
    if (!h) return 0;
    if (index < 0) return h->s.lengths.size();
    return index < static_cast<int>(h->s.codes.size()) ? h->s.codes[index].size() : 0;
}

mpc_status mpc_streams_copy(const mpc_streams* h, int index, uint16_t* dst) {
    if (!h || !dst) return fail(MPC_ERR_ARGUMENT, "null argument");
    const std::vector<uint16_t>* v = nullptr;
    if (index < 0) v = &h->s.lengths;
    else if (index < static_cast<int>(h->s.codes.size())) v = &h->s.codes[index];
    if (!v) return fail(MPC_ERR_ARGUMENT, "stream index %d out of range", index);
    if (!v->empty()) std::memcpy(dst, v->data(), v->size() * 2);
    return MPC_OK;
}

void mpc_streams_free(mpc_streams* h) { delete h; }

mpc_status mpc_huffman_encode(const uint16_t* data, size_t n, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if ((!data && n) || !bytes || !nbytes) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc::BitWriter w;
    mpc::huffman_encode(data, n, w);
    *bytes = give_bytes(w.bytes(), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_huffman_decode(const uint8_t* bytes, size_t nbytes, uint16_t** data, size_t* n) {
    return guarded([&]() -> mpc_status {
    if (!bytes || !data || !n) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc::BitReader r(bytes, nbytes);
    std::vector<uint16_t> out;
    if (!mpc::huffman_decode(r, out)) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
    *data = give_u16(out, n);
    return *data ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_rle_encode(const uint16_t* data, size_t n, uint16_t** out, size_t* n_out) {
    return guarded([&]() -> mpc_status {
    if ((!data && n) || !out || !n_out) return fail(MPC_ERR_ARGUMENT, "null argument");
    *out = give_u16(mpc::rle_encode(data, n), n_out);
    return *out ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_rle_decode(const uint16_t* data, size_t n, uint16_t** out, size_t* n_out) {
    return guarded([&]() -> mpc_status {
    if ((!data && n) || !out || !n_out) return fail(MPC_ERR_ARGUMENT, "null argument");
    *out = give_u16(mpc::rle_decode(data, n), n_out);
    return *out ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

// ---- bit-level primitives of CompressionLib/inc/BitBuffer.h, as the entropy stage uses them ----
uint32_t mpc_zigzag_encode(int32_t x) { return mpc::zigzag_encode(x); }
int32_t mpc_zigzag_decode(uint32_t x) { return mpc::zigzag_decode(x); }
uint32_t mpc_golomb_length(uint32_t value, uint32_t m) { return m ? mpc::golomb_length(value, m) : 0; }
uint32_t mpc_elias_fano_length(size_t n, uint16_t max_symbol) { return mpc::elias_fano_length(n, max_symbol); }

mpc_status mpc_bits_pack(const uint64_t* values, const int* widths, size_t n, uint8_t** bytes, size_t* nbytes, size_t* nbits) {
    return guarded([&]() -> mpc_status {
    if ((!values || !widths) && n) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (!bytes || !nbytes || !nbits) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc::BitWriter w;
    for (size_t i = 0; i < n; ++i) {
        if (widths[i] < 0 || widths[i] > 64) return fail(MPC_ERR_ARGUMENT, "Invalid bit width");      // BitBuffer.cpp:80 throws here
        w.put(values[i], widths[i]);
    }
    *nbits = w.bit_size();
    *bytes = give_bytes(w.bytes(), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_bits_unpack(const uint8_t* bytes, size_t nbytes, const int* widths, size_t n, uint64_t* values, size_t* remaining_bits) {
    if ((!bytes && nbytes) || ((!widths || !values) && n)) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc::BitReader r(bytes, nbytes);
    for (size_t i = 0; i < n; ++i) {
        if (widths[i] < 0 || widths[i] > 64) return fail(MPC_ERR_ARGUMENT, "Invalid bit width");
        values[i] = r.get(widths[i]);
    }
    if (remaining_bits) *remaining_bits = r.remaining();
    return MPC_OK;
}

mpc_status mpc_golomb_encode(const uint32_t* values, size_t n, uint32_t m, uint8_t** bytes, size_t* nbytes, size_t* nbits) {
    return guarded([&]() -> mpc_status {
    if ((!values && n) || !bytes || !nbytes || !nbits || m == 0) return fail(MPC_ERR_ARGUMENT, "bad argument");
    mpc::BitWriter w;
    for (size_t i = 0; i < n; ++i) mpc::golomb_write(values[i], m, w);
    *nbits = w.bit_size();
    *bytes = give_bytes(w.bytes(), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_golomb_decode(const uint8_t* bytes, size_t nbytes, size_t n, uint32_t m, uint32_t* values, size_t* remaining_bits) {
    if ((!bytes && nbytes) || (!values && n) || m == 0) return fail(MPC_ERR_ARGUMENT, "bad argument");
    mpc::BitReader r(bytes, nbytes);
    for (size_t i = 0; i < n; ++i) values[i] = mpc::golomb_read(m, r);
    if (remaining_bits) *remaining_bits = r.remaining();
    return MPC_OK;
}

mpc_status mpc_elias_fano_encode(const uint16_t* sorted, size_t n, uint16_t max_symbol, uint8_t** bytes, size_t* nbytes, size_t* nbits) {
    return guarded([&]() -> mpc_status {
    if ((!sorted && n) || !bytes || !nbytes || !nbits) return fail(MPC_ERR_ARGUMENT, "null argument");
    for (size_t i = 0; i < n; ++i)
        if (sorted[i] > max_symbol || (i && sorted[i] < sorted[i - 1])) return fail(MPC_ERR_ARGUMENT, "sequence not sorted or beyond max_symbol");
    mpc::BitWriter w;
    mpc::elias_fano_write(sorted, n, max_symbol, w);
    *nbits = w.bit_size();
    *bytes = give_bytes(w.bytes(), nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_elias_fano_decode(const uint8_t* bytes, size_t nbytes, size_t n, uint16_t max_symbol, uint16_t* sorted, size_t* remaining_bits) {
    if ((!bytes && nbytes) || (!sorted && n)) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc::BitReader r(bytes, nbytes);
    if (!mpc::elias_fano_read(sorted, n, max_symbol, r)) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
    if (remaining_bits) *remaining_bits = r.remaining();
    return MPC_OK;
}

// ------------------------------------------------------------------------------------------------
// Entropy stage with the per-symbol work on the device (mp_entropy.hip).  Phase 1 is enqueued behind the stream assembly;
// a worker then fetches the per-stream statistics, builds the tables (host_bitstream.cpp: plan_stream), sends them back and
// runs phase 2, which writes the codes into the container on the device; only the finished bytes cross PCIe.
// MPC_HOST_ENTROPY=1 keeps the whole stage on the host (the symbols cross instead); the same route is taken when a stream
// is outside what the device tables hold (more distinct symbols than the triple list, a code longer than 32 bits).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr unsigned kTripleCap = 1u << 20;

// MPC_ENTROPY_TRIPLES=n lowers the number of (symbol, count, first position) triples a frame may have before it takes the host
// route (tests use it to exercise that route behind a completed phase 1)
unsigned triple_limit() {
    static const unsigned limit = [] {
        const int v = env_int("MPC_ENTROPY_TRIPLES", 0);
        return v > 0 && static_cast<unsigned>(v) < kTripleCap ? static_cast<unsigned>(v) : kTripleCap;
    }();
    return limit;
}

bool host_entropy_forced() {
    static const bool forced = env_int("MPC_HOST_ENTROPY", 0) != 0;
    return forced;
}

struct EntropyBuffers {
    mpc::EntropyArgs args{};
    uint8_t* d_out = nullptr;
    size_t out_capacity = 0;             // bytes, device and host
    unsigned long long capacity_symbols = 0;
    // pinned
    mpc::EntStream* h_streams = nullptr;
    unsigned* h_totals = nullptr;
    unsigned* h_triples = nullptr;
    unsigned* h_entries = nullptr;
    uint8_t* h_out = nullptr;
};

// carve (and grow) slot `sl`'s buffers for frames of `tiles` tiles; the symbols of the streams live in the caller's buffers
mpc_status entropy_buffers(mpc_context::EntropySlot& e, size_t tiles, int K, EntropyBuffers* b) {
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const int S = 6 * K + 1;
    const size_t n_tc = 3 * tiles;
    const unsigned long long cap_symbols = n_tc + 2ULL * n_tc * K;
    const size_t blocks = mpc::entropy_max_blocks(cap_symbols, S);
    const size_t streams_b = up(sizeof(mpc::EntStream) * S), totals_b = up(4 * sizeof(unsigned));
    const size_t blk_u32 = up(sizeof(unsigned) * blocks), blk_u64 = up(sizeof(unsigned long long) * blocks);
    const size_t triples_b = up(sizeof(unsigned) * 3 * kTripleCap);
    const size_t packed_b = up(sizeof(uint16_t) * 2 * n_tc * K);
    const size_t tcode_b = up(sizeof(unsigned) * 65536 * S), tlen_b = up(65536 * static_cast<size_t>(S));
    const size_t out_b = up(packed_b + 65536);
    const size_t hist_b = tcode_b;                        // ghist, gfirst: [S][65536] words each
    const size_t dev_need = streams_b + totals_b + 7 * blk_u32 + blk_u64 + packed_b + tcode_b + tlen_b + out_b + 2 * hist_b;
    const size_t host_need = streams_b + totals_b + 2 * triples_b + out_b;
    if (dev_need > e.dev_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (e.dev) (void)hipFree(e.dev);
        e.dev = nullptr;
        e.dev_bytes = 0;
        const hipError_t err = hipMalloc(&e.dev, dev_need);
        if (err != hipSuccess) return fail(MPC_ERR_ALLOC, "entropy stage buffers of %zu bytes: %s", dev_need, hipGetErrorString(err));
        e.dev_bytes = dev_need;
        e.tiles = 0;
    }
    if (host_need > e.host_bytes) {
        if (e.host) (void)hipHostFree(e.host);
        e.host = nullptr;
        e.host_bytes = 0;
        const hipError_t err = hipHostMalloc(&e.host, host_need, hipHostMallocMapped);
        if (err != hipSuccess) return fail(MPC_ERR_ALLOC, "pinned entropy stage buffers of %zu bytes: %s", host_need, hipGetErrorString(err));
        e.host_bytes = host_need;
    }
    // the kernels read and write the small host-side tables in place (mapped, coherent host memory)
    char* mapped = nullptr;
    HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&mapped), e.host, 0));
    char* d = static_cast<char*>(e.dev);
    char* h = static_cast<char*>(e.host);
    mpc::EntropyArgs& a = b->args;
    a = mpc::EntropyArgs{};
    a.n_lengths = static_cast<unsigned>(n_tc);
    a.n_streams = S;
    a.streams = reinterpret_cast<mpc::EntStream*>(d); d += streams_b;
    a.totals = reinterpret_cast<unsigned*>(d); d += totals_b;
    a.blk_stream = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_lead = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_inner = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_tail = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_carry = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_out = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_bits = reinterpret_cast<unsigned*>(d); d += blk_u32;
    a.blk_bit_off = reinterpret_cast<unsigned long long*>(d); d += blk_u64;
    a.packed = reinterpret_cast<uint16_t*>(d); d += packed_b;
    a.tcode = reinterpret_cast<unsigned*>(d); d += tcode_b;
    a.tlen = reinterpret_cast<uint8_t*>(d); d += tlen_b;
    b->d_out = reinterpret_cast<uint8_t*>(d); d += out_b;
    a.triple_cap = kTripleCap;
    a.ghist = reinterpret_cast<unsigned*>(d); d += hist_b;
    a.gfirst = reinterpret_cast<unsigned*>(d); d += hist_b;     // the last hist_b bytes of the allocation
    a.out32 = reinterpret_cast<unsigned*>(b->d_out);
    if (e.tiles != tiles || e.K != K) {
        // the dense code tables and the histogram are zero between frames (the kernels clear what they set), the first
        // positions all ones; a slot carved for another geometry holds them elsewhere
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(a.tcode, 0, tcode_b));
        HIP_TRY(hipMemset(a.tlen, 0, tlen_b));
        HIP_TRY(hipMemset(a.ghist, 0, hist_b));
        HIP_TRY(hipMemset(a.gfirst, 0xFF, hist_b));
        e.tiles = tiles;
        e.K = K;
    }
    b->out_capacity = out_b;
    b->capacity_symbols = cap_symbols;
    b->h_streams = reinterpret_cast<mpc::EntStream*>(h); a.host_streams = reinterpret_cast<mpc::EntStream*>(mapped + (h - static_cast<char*>(e.host))); h += streams_b;
    b->h_totals = reinterpret_cast<unsigned*>(h); a.host_totals = reinterpret_cast<unsigned*>(mapped + (h - static_cast<char*>(e.host))); h += totals_b;
    b->h_triples = reinterpret_cast<unsigned*>(h); a.triples = reinterpret_cast<unsigned*>(mapped + (h - static_cast<char*>(e.host))); h += triples_b;
    b->h_entries = reinterpret_cast<unsigned*>(h); a.entries = reinterpret_cast<const unsigned*>(mapped + (h - static_cast<char*>(e.host))); h += triples_b;
    b->h_out = reinterpret_cast<uint8_t*>(h); h += out_b;
    return MPC_OK;
}

enum class EntropyResult { kDone, kNeedsHost, kFailed };

// Wait for an event.  `spin`: poll it (a single frame's latency is a chain of such waits, and a sleeping thread takes tens of
// microseconds to come back); otherwise let the thread sleep -- in the frame pipeline the table building wants the cores.
hipError_t wait_event(hipEvent_t ev, bool spin) {
    if (!spin) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        __builtin_ia32_pause();
    }
}

double trace_ms() {
    static const auto origin = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - origin).count();
}

// The host's part and phase 2, in two steps.  Phase 1 has completed (the caller waited for an event behind it): the
// statistics are in the slot's host mirrors.
//   entropy_tables   builds the code tables and enqueues on `s`, in order: table import, the code kernels, the container's way
//                    to the host, `done`.  kNeedsHost: nothing enqueued, take the host route.
//   entropy_collect  waits for `done`, checks the device's bit counts against the tables', patches the host's pieces in.
struct EntropyPending {
    std::vector<mpc::StreamPlan> plans;
    mpc::BitWriter head;
    size_t total_bytes = 0;
};

EntropyResult entropy_tables(const EntropyBuffers& b, int device_block_size, int width, int height, int K, const double* quant,
                             hipStream_t s, hipEvent_t done, EntropyPending* pending, double* stamps = nullptr) {
    const mpc::EntropyArgs& a = b.args;
    auto stamp = [&](int i) { if (stamps) stamps[i] = trace_ms(); };
    const int S = a.n_streams;
    if (b.h_totals[3] != 0 || b.h_totals[2] > triple_limit()) return EntropyResult::kNeedsHost;
    stamp(0);
    std::vector<mpc::StreamPlan>& plans = pending->plans;
    plans.assign(static_cast<size_t>(S), mpc::StreamPlan());
    std::vector<int> order(static_cast<size_t>(S));             // the streams with the most symbols to build a tree from first
    for (int j = 0; j < S; ++j) order[static_cast<size_t>(j)] = j;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return b.h_streams[x].distinct > b.h_streams[y].distinct; });
    mpc::parallel_jobs(S, [&](int job) {
        const int j = order[static_cast<size_t>(job)];
        const mpc::EntStream& st = b.h_streams[j];
        mpc::plan_stream(j != 0, st.shorter != 0, st.rle_size, st.eff_n, st.largest, b.h_triples + 3 * static_cast<size_t>(st.triple_off),
                         st.distinct, plans[static_cast<size_t>(j)]);
    });
    pending->head = mpc::container_head(width, height, K, device_block_size, quant);
    unsigned long long bit = pending->head.bit_size(), raw_symbols = 0;
    size_t n_entries = 0;
    for (int j = 0; j < S; ++j) {
        const mpc::StreamPlan& p = plans[static_cast<size_t>(j)];
        if (p.mode == 0 && p.max_code_length > 32) return EntropyResult::kNeedsHost;
        mpc::EntStream& st = b.h_streams[j];
        bit += p.pre.bit_size();
        st.bit_off = bit;
        st.mode = static_cast<unsigned>(p.mode);
        st.m = p.m;
        bit += p.payload_bits + p.post.bit_size();
        raw_symbols += st.n;
        n_entries += p.entries.size() / 3;
    }
    const size_t total_bytes = static_cast<size_t>((bit + 7) / 8), out_words = (total_bytes + 3) / 4;
    if (out_words * 4 > b.out_capacity || n_entries > kTripleCap) return EntropyResult::kNeedsHost;
    pending->total_bytes = total_bytes;
    size_t at = 0;
    for (int j = 0; j < S; ++j) {
        const std::vector<uint32_t>& e = plans[static_cast<size_t>(j)].entries;
        for (size_t k = 0; k < e.size(); k += 3) {
            b.h_entries[at++] = (static_cast<unsigned>(j) << 16) | e[k];
            b.h_entries[at++] = e[k + 1];
            b.h_entries[at++] = e[k + 2];
        }
    }
    stamp(1);                                                   // tables built
    mpc::EntropyArgs a2 = a;
    a2.n_entries = static_cast<unsigned>(n_entries);
    a2.out_words = out_words;
    const bool ok = hipMemsetAsync(b.d_out, 0, out_words * 4, s) == hipSuccess && mpc::launch_entropy_phase2(a2, raw_symbols, s) == 0 &&
                    hipMemcpyAsync(b.h_out, b.d_out, out_words * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
                    hipEventRecord(done, s) == hipSuccess;
    return ok ? EntropyResult::kDone : EntropyResult::kFailed;
}

EntropyResult entropy_collect(const EntropyBuffers& b, const EntropyPending& pending, hipEvent_t done, uint8_t** blob, size_t* nbytes,
                              double* stamps = nullptr, bool spin = false) {
    if (wait_event(done, spin) != hipSuccess) return EntropyResult::kFailed;
    if (stamps) stamps[2] = trace_ms();                         // codes written, bytes on the host
    const int S = b.args.n_streams;
    for (int j = 0; j < S; ++j)                                 // the device wrote exactly the bits the tables promise
        if (b.h_streams[j].coded_bits != pending.plans[static_cast<size_t>(j)].payload_bits) return EntropyResult::kFailed;
    mpc::or_bits(b.h_out, b.out_capacity, 0, pending.head);
    for (int j = 0; j < S; ++j) {
        const mpc::StreamPlan& p = pending.plans[static_cast<size_t>(j)];
        const unsigned long long payload = b.h_streams[j].bit_off;
        mpc::or_bits(b.h_out, b.out_capacity, static_cast<size_t>(payload - p.pre.bit_size()), p.pre);
        mpc::or_bits(b.h_out, b.out_capacity, static_cast<size_t>(payload + p.payload_bits), p.post);
    }
    uint8_t* out = static_cast<uint8_t*>(std::malloc(pending.total_bytes ? pending.total_bytes : 1));
    if (!out) return EntropyResult::kFailed;
    {   // fresh pages: a few threads fault them in and copy
        const size_t total = pending.total_bytes, piece = ((total + 7) / 8 + 4095) & ~static_cast<size_t>(4095);
        const uint8_t* src = b.h_out;
        mpc::parallel_jobs(total > (1u << 20) ? 8 : 1, [&](int k) {
            const size_t lo = std::min(total, piece * static_cast<size_t>(k));
            const size_t hi = total > (1u << 20) ? std::min(total, lo + piece) : total;
            if (hi > lo) std::memcpy(out + lo, src + lo, hi - lo);
        });
    }
    *blob = out;
    *nbytes = pending.total_bytes;
    return EntropyResult::kDone;
}

// both steps; `enqueued()` (if any) is called between them: once phase 2 is on the stream, or once it is clear that it will not be
EntropyResult finish_entropy_on_device(const EntropyBuffers& b, int device_block_size, int width, int height, int K, const double* quant,
                                       hipStream_t s, hipEvent_t done, const std::function<void()>& enqueued, uint8_t** blob,
                                       size_t* nbytes, double* stamps = nullptr, bool spin = false) {
    EntropyPending pending;
    const EntropyResult r = entropy_tables(b, device_block_size, width, height, K, quant, s, done, &pending, stamps);
    if (enqueued) enqueued();
    if (r != EntropyResult::kDone) return r;
    return entropy_collect(b, pending, done, blob, nbytes, stamps, spin);
}
}  // namespace

// compressed::encodeImage for a sequence of equally sized frames: device tile encode, device stream assembly (only the live
// symbols cross PCIe, stream by stream), host entropy stage -- pipelined over kSeqSlots slots.  Frames come from host memory
// (uploaded through the slot's pinned image on an upload stream) or are already resident on the device.
static mpc_status encode_sequence(mpc_context* c, const uint8_t* const* frames, bool on_device, int n_frames, int width, int height,
                                  const double* quant, uint8_t** bytes, size_t* nbytes) {
    if (!c || !frames || !bytes || !nbytes || n_frames < 1) return fail(MPC_ERR_ARGUMENT, "bad argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    if (width < 1 || height < 1) return fail(MPC_ERR_ARGUMENT, "bad geometry");
    for (int f = 0; f < n_frames; ++f) {
        if (!frames[f]) return fail(MPC_ERR_ARGUMENT, "null frame");
        bytes[f] = nullptr;
        nbytes[f] = 0;
    }
    const int tiles_y = (height + 7) / 8;
    const size_t tiles = static_cast<size_t>((width + 7) / 8) * tiles_y;
    const size_t n_tc = tiles * 3;
    const int K = c->K;
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    HIP_TRY(hipSetDevice(c->device));
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t img_bytes = static_cast<size_t>(3) * width * height;
    const size_t counts_bytes = up(sizeof(uint16_t) * n_tc), choices_bytes = up(sizeof(mpc_basis_choice) * n_tc * K);
    const size_t off_bytes = up(sizeof(unsigned long long) * (6 * static_cast<size_t>(K) + 1));
    const size_t symbols_bytes = up(sizeof(uint16_t) * 2 * n_tc * K);            // worst case: every record alive
    const size_t live_bytes = up(sizeof(unsigned) * mpc::stream_workspace_words(static_cast<long long>(tiles), K));
    const size_t sizes_bytes = up(sizeof(unsigned) * 3 * K), dc_bytes = up(sizeof(uint16_t) * n_tc);
    // host slot: image | counts | stream offsets | symbols;  device slot: image | counts | records | block_live | sizes | offsets | symbols | dc
    const size_t host_slot = (on_device ? 0 : up(img_bytes)) + counts_bytes + off_bytes + symbols_bytes;
    const size_t dev_slot = (on_device ? 0 : up(img_bytes)) + counts_bytes + choices_bytes + live_bytes + sizes_bytes + off_bytes + symbols_bytes + dc_bytes;
    constexpr size_t S = mpc_context::kSeqSlots;
    const size_t slots = std::min<size_t>(S, static_cast<size_t>(n_frames));
    // a sequence gets every slot's buffers at once: a later, longer call then finds them (allocating pinned memory takes
    // tens of milliseconds)
    const size_t alloc_slots = n_frames > 1 ? S : 1;
    if (alloc_slots * host_slot > c->host_stage_bytes) {
        if (c->host_stage) (void)hipHostFree(c->host_stage);
        c->host_stage = nullptr;
        c->host_stage_bytes = 0;
        const hipError_t e = hipHostMalloc(&c->host_stage, alloc_slots * host_slot, hipHostMallocDefault);
        if (e != hipSuccess) return fail(MPC_ERR_ALLOC, "pinned staging of %zu bytes: %s", alloc_slots * host_slot, hipGetErrorString(e));
        c->host_stage_bytes = alloc_slots * host_slot;
    }
    if (alloc_slots * dev_slot > c->stage_bytes) {
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_bytes = 0;
        const hipError_t e = hipMalloc(&c->stage, alloc_slots * dev_slot);
        if (e != hipSuccess) return fail(MPC_ERR_ALLOC, "device staging of %zu bytes: %s", alloc_slots * dev_slot, hipGetErrorString(e));
        c->stage_bytes = alloc_slots * dev_slot;
    }
    if (!c->seq_up) {
        // The side streams at the highest priority, the pursuits' at the lowest: CUs a pursuit gives up at its end go to the
        // waiting chains of small kernels before the next pursuit's workgroups (MPC_SIDE_PRIORITY=0: all equal).  Measured at
        // 4928x3264 (tools/ab_env_bench.sh): 4 740 against 4 660 Mpix/s, with two side streams for all slots (below).
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const bool prio = env_int("MPC_SIDE_PRIORITY", 1) != 0 && least != greatest;
        c->seq_prioritised = prio;
        HIP_TRY(hipStreamCreateWithFlags(&c->seq_up, hipStreamNonBlocking));
        HIP_TRY(prio ? hipStreamCreateWithPriority(&c->seq_compute, hipStreamNonBlocking, least) : hipStreamCreateWithFlags(&c->seq_compute, hipStreamNonBlocking));
        for (auto& s : c->seq_down)
            HIP_TRY(prio ? hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        for (hipEvent_t& e : c->seq_pursuit_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t& e : c->seq_stripe_up) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        // blocking events: a thread waiting for the device sleeps instead of spinning (the entropy stage wants the cores)
        for (auto& slot : c->seq_events)
            for (hipEvent_t& e : slot) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
    }
    HIP_TRY(ensure_workspace(c, static_cast<long long>(n_tc)) == MPC_OK ? hipSuccess : hipErrorOutOfMemory);
    const double* q = quant ? quant : c->quant.data();
    const bool device_entropy = !host_entropy_forced();
    // a single frame: nothing to overlap with, so its worker runs on the calling thread (no thread to start and to join) and
    // polls the device instead of sleeping
    const bool single = n_frames == 1;
    // A pursuit on every CU leaves the small kernels behind the previous frames' pursuits (stream assembly, entropy phases: ~8 %
    // of a frame's CU time) nowhere to run but the gap between two pursuits, where they are latency-bound and the chip idles for
    // ~0.45 ms per 16 Mpixel frame.  With one CU in eight left free they run beside the pursuit instead.  Measured on one box
    // (tools/ab_env_bench.sh, MPC_SEQ_WORKGROUPS = pursuit workgroups of 256): 4928x3264 K=32  256: 4 780, 240: 4 620, 224: 5 060,
    // 216: 4 980, 208: 4 870 Mpix/s; 1920x1080 K=8  256: 3 300 - 3 790, 224: 4 150; 7680x4320 K=16  256: 5 570, 224: 5 860 (with
    // 16 CUs the chains cannot keep up and the pursuits wait for them).
    struct SeqWorkgroups {
        mpc_context* c;
        ~SeqWorkgroups() { c->seq_workgroups = 0; }
    } seq_workgroups{c};
    if (!single && c->num_cus >= 16) {
        static const int env_wg = env_int("MPC_SEQ_WORKGROUPS", -1);
        c->seq_workgroups = env_wg >= 0 ? env_wg : c->num_cus - c->num_cus / 8;
    }
    // The pipeline's streams.  `seq_compute`: the pursuits, one behind the other.  Side stream A: behind pursuit(f) (an event) the
    // stream assembly and entropy phase 1 of frame f.  Side stream B: phase 2 and the container's copy of frame f, enqueued by the
    // frame's worker once it has built the code tables from phase 1's statistics (mapped host memory written by the kernels
    // themselves; the host waits on events only).  Pursuit(f) waits (events) for the assembly + phase 1 of frame f - 2 and for
    // the phase 2 of frame f - 3, so nothing piles up; with the CUs the pursuits leave free (above) both chains run beside the
    // pursuits of the following frames.  The shapes this replaced -- everything on one ordered queue; phase 2 alone on a side
    // stream -- are in DESIGN.md 4 and 9.
    hipStream_t pursuit_stream = c->seq_compute;
    std::future<void> phase2_enqueued[S];
    EntropyBuffers ent[S];
    if (device_entropy)
        for (size_t sl = 0; sl < alloc_slots; ++sl) {
            const mpc_status es = entropy_buffers(c->ent[sl], tiles, K, &ent[sl]);
            if (es != MPC_OK) return es;
        }
    struct Pending {
        std::future<std::pair<uint8_t*, size_t>> result;     // malloc'ed container, or {nullptr, 0}
        int frame = -1;
    } pending[S];
    mpc_status st = MPC_OK;
    auto collect = [&](Pending& p) {
        if (p.frame < 0) return;
        const std::pair<uint8_t*, size_t> blob = p.result.get();
        if (st == MPC_OK && !blob.first) st = fail(MPC_ERR_HIP, "record download or container allocation failed");
        if (st == MPC_OK) {
            bytes[p.frame] = blob.first;
            nbytes[p.frame] = blob.second;
        } else {
            std::free(blob.first);
        }
        p.frame = -1;
    };
#define MPC_SEQ_TRY(call)                                                                         \
    {                                                                                             \
        const hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { st = fail(MPC_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); break; } \
    }
    // Host frames: frame g is copied into its slot's pinned image by a few threads (the caller's memory is pageable: the runtime
    // would stage it through one thread at a few GB/s) and sent to the device while frame g - 1 is still being enqueued and
    // encoded; its pursuit waits for the event behind the copy.
    std::future<hipError_t> uploads[S];
    auto start_upload = [&](int g) {
        const int usl = g % static_cast<int>(slots);
        collect(pending[usl]);            // the slot's previous frame is through
        uint8_t* d_img = reinterpret_cast<uint8_t*>(static_cast<char*>(c->stage) + static_cast<size_t>(usl) * dev_slot);
        uint8_t* pinned_rgb = reinterpret_cast<uint8_t*>(static_cast<char*>(c->host_stage) + static_cast<size_t>(usl) * host_slot);
        const uint8_t* src = frames[g];
        hipEvent_t ev_up = c->seq_events[usl][0];
        hipStream_t up_stream = c->seq_up;
        const int device = c->device;
        uploads[usl] = std::async(single ? std::launch::deferred : std::launch::async, [=]() -> hipError_t {
            // In chunks: a chunk goes to the device as soon as it is in pinned memory, so the DMA of the first chunks runs while
            // the later ones are still being copied (a 16 Mpixel frame: 48 MB staged at memcpy speed, then sent at PCIe speed).
            static const size_t parts = static_cast<size_t>(std::min(64, std::max(1, env_int("MPC_UPLOAD_CHUNKS", 32))));
            const size_t chunk = std::max<size_t>(size_t(1) << 20, (((img_bytes + parts - 1) / parts) + 4095) & ~static_cast<size_t>(4095));
            const int chunks = static_cast<int>((img_bytes + chunk - 1) / chunk);
            std::atomic<int> failed{static_cast<int>(hipSuccess)};
            mpc::parallel_io_jobs(chunks, 8, [&](int k) {
                const size_t lo = chunk * static_cast<size_t>(k), hi = std::min(img_bytes, lo + chunk);
                std::memcpy(pinned_rgb + lo, src + lo, hi - lo);
                hipError_t e = hipSetDevice(device);
                if (e == hipSuccess) e = hipMemcpyAsync(d_img + lo, pinned_rgb + lo, hi - lo, hipMemcpyHostToDevice, up_stream);
                if (e != hipSuccess) failed.store(static_cast<int>(e));
            });
            hipError_t e = static_cast<hipError_t>(failed.load());
            if (e == hipSuccess) e = hipSetDevice(device);
            if (e == hipSuccess) e = hipEventRecord(ev_up, up_stream);
            return e;
        });
    };
    // a single host frame in row stripes (below): MPC_SINGLE_STRIPES, 1 = whole frame at once; needs the persistent kernel (the
    // step-synchronous cross-check path writes stripe order only)
    const int stripes_env = std::min<int>(mpc_context::kSingleStripes, std::max(1, env_int("MPC_SINGLE_STRIPES", 0)));
    // measured (tools/single_frame_trace.py, host RGB -> bytes): 4928x3264  1 / 2 / 3 / 4 stripes: 5.69 / 5.25 / 4.92 / 5.20 ms;
    // 1920x1080: 1.24 / 1.42 / 1.45 / 1.63 ms -- every further launch costs a prologue (144 KiB of LDS per workgroup) and a tail,
    // more than a 6 MB copy takes
    // 7680x4320: 9.5 / 8.3 / 8.1 ms with 1 / 3 / 4
    const int single_stripes = env_int("MPC_SINGLE_STRIPES", 0) > 0 ? stripes_env : (img_bytes >= (size_t(80) << 20) ? 4 : (img_bytes >= (size_t(32) << 20) ? 3 : 1));
    const bool striped_single = single && !on_device && single_stripes > 1 && tiles_y >= 4 * single_stripes && !steps_path();
    if (!on_device && !striped_single) start_upload(0);
    for (int f = 0; f < n_frames && st == MPC_OK; ++f) {
        const int sl = f % static_cast<int>(slots);
        Pending& slot = pending[sl];
        collect(slot);                    // frame f - slots is done with this slot: its download and its entropy stage have finished
        if (st != MPC_OK) break;
        // how far the pursuits may run ahead of the small kernels behind them: pursuit(f) waits for the stream assembly + phase 1
        // of frame f - lag_assembly and for the phase 2 of frame f - back.  Measured (tools/ab_env_bench.sh, 4928x3264) while the
        // pursuits still filled every CU: lags 2 / 3 -> 4 660 Mpix/s, 3 / 3 -> 4 610 - 4 690, 3 / 4 -> 4 000 - 4 300, 4 / 5 -> 4 090 -
        // 4 680 (more slack let the chains of several frames pile up in front of one pursuit's end); with CUs left free for the
        // chains 2 / 2, 2 / 3 and 3 / 4 are within 1 % of each other.
        static const int lag_assembly = std::min(4, std::max(2, env_int("MPC_LAG_ASSEMBLY", 2)));
        static const int lag_phase2 = std::min(5, std::max(lag_assembly, env_int("MPC_LAG_PHASE2", 3)));
        const int back = lag_phase2;
        if (f >= back && phase2_enqueued[(f - back) % static_cast<int>(slots)].valid()) {
            // frame f - back's phase 2 is on its slot's stream by now: this frame's pursuit starts behind it (the event is the one
            // its worker recorded behind the container's copy; on the host route it is an old one and the wait is empty)
            phase2_enqueued[(f - back) % static_cast<int>(slots)].get();
            MPC_SEQ_TRY(hipStreamWaitEvent(pursuit_stream, c->seq_events[(f - back) % static_cast<int>(slots)][2], 0));
        }
        if (f >= lag_assembly)
            MPC_SEQ_TRY(hipStreamWaitEvent(pursuit_stream, c->seq_events[(f - lag_assembly) % static_cast<int>(slots)][1], 0));
        char* dbase = static_cast<char*>(c->stage) + static_cast<size_t>(sl) * dev_slot;
        char* hbase = static_cast<char*>(c->host_stage) + static_cast<size_t>(sl) * host_slot;
        const uint8_t* d_rgb = frames[f];
        bool encoded_in_stripes = false;
        if (!on_device && !striped_single) {
            uint8_t* d_img = reinterpret_cast<uint8_t*>(dbase);
            dbase += up(img_bytes);
            hbase += up(img_bytes);
            MPC_SEQ_TRY(uploads[sl].get());
            MPC_SEQ_TRY(hipStreamWaitEvent(pursuit_stream, c->seq_events[sl][0], 0));
            if (f + 1 < n_frames) start_upload(f + 1);
            if (st != MPC_OK) break;
            d_rgb = d_img;
        }
        if (striped_single) {
            // One frame from host memory: nothing to overlap its upload with but its own tile encode.  The frame goes up in row
            // stripes and each stripe's tile encode starts behind its own copy (an event), writing its records where one launch
            // over the whole frame would put them: the copy of stripe s + 1 runs beside the encode of stripe s.
            uint8_t* d_img = reinterpret_cast<uint8_t*>(dbase);
            uint8_t* pinned_rgb = reinterpret_cast<uint8_t*>(hbase);
            dbase += up(img_bytes);
            hbase += up(img_bytes);
            uint16_t* d_counts_s = reinterpret_cast<uint16_t*>(dbase);
            mpc_basis_choice* d_choices_s = reinterpret_cast<mpc_basis_choice*>(dbase + counts_bytes);
            MPC_SEQ_TRY(hipMemsetAsync(d_choices_s, 0, sizeof(mpc_basis_choice) * n_tc * K, pursuit_stream));
            const uint8_t* src = frames[f];
            const size_t row_bytes = static_cast<size_t>(3) * width;
            for (int sp = 0; sp < single_stripes && st == MPC_OK; ++sp) {
                const int rb = static_cast<int>(static_cast<long long>(tiles_y) * sp / single_stripes);
                const int re = static_cast<int>(static_cast<long long>(tiles_y) * (sp + 1) / single_stripes);
                const size_t lo_b = row_bytes * static_cast<size_t>(8 * rb), hi_b = row_bytes * static_cast<size_t>(std::min(height, 8 * re));
                const size_t chunk = std::max<size_t>(size_t(1) << 20, (((hi_b - lo_b + 7) / 8) + 4095) & ~static_cast<size_t>(4095));
                const int chunks = static_cast<int>((hi_b - lo_b + chunk - 1) / chunk);
                std::atomic<int> failed{static_cast<int>(hipSuccess)};
                const int device = c->device;
                hipStream_t up_stream = c->seq_up;
                mpc::parallel_io_jobs(chunks, 8, [&](int k) {
                    const size_t lo = lo_b + chunk * static_cast<size_t>(k), hi = std::min(hi_b, lo + chunk);
                    std::memcpy(pinned_rgb + lo, src + lo, hi - lo);
                    hipError_t e = hipSetDevice(device);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_img + lo, pinned_rgb + lo, hi - lo, hipMemcpyHostToDevice, up_stream);
                    if (e != hipSuccess) failed.store(static_cast<int>(e));
                });
                MPC_SEQ_TRY(static_cast<hipError_t>(failed.load()));
                MPC_SEQ_TRY(hipEventRecord(c->seq_stripe_up[sp], up_stream));
                MPC_SEQ_TRY(hipStreamWaitEvent(pursuit_stream, c->seq_stripe_up[sp], 0));
                st = encode_batch_device(c, d_img, 1, 0, width, height, row_bytes, rb, re, quant, d_counts_s, d_choices_s, nullptr, nullptr,
                                         pursuit_stream, true);
            }
            if (st != MPC_OK) break;
            d_rgb = d_img;
            encoded_in_stripes = true;
        }
        uint16_t* d_counts = reinterpret_cast<uint16_t*>(dbase);
        mpc_basis_choice* d_choices = reinterpret_cast<mpc_basis_choice*>(dbase + counts_bytes);
        mpc::StreamArgs sa{};
        sa.counts = d_counts;
        sa.choices = reinterpret_cast<const uint32_t*>(d_choices);
        sa.tiles = static_cast<long long>(tiles);
        sa.K = K;
        sa.block_live = reinterpret_cast<unsigned*>(dbase + counts_bytes + choices_bytes);
        sa.sizes = reinterpret_cast<unsigned*>(dbase + counts_bytes + choices_bytes + live_bytes);
        sa.stream_off = reinterpret_cast<unsigned long long*>(dbase + counts_bytes + choices_bytes + live_bytes + sizes_bytes);
        sa.symbols = reinterpret_cast<uint16_t*>(dbase + counts_bytes + choices_bytes + live_bytes + sizes_bytes + off_bytes);
        sa.dc_tmp = reinterpret_cast<uint16_t*>(dbase + counts_bytes + choices_bytes + live_bytes + sizes_bytes + off_bytes + symbols_bytes);
        uint16_t* counts = reinterpret_cast<uint16_t*>(hbase);
        unsigned long long* off = reinterpret_cast<unsigned long long*>(hbase + counts_bytes);
        uint16_t* symbols = reinterpret_cast<uint16_t*>(hbase + counts_bytes + off_bytes);
        if (!encoded_in_stripes)
            st = mpc_encode_tiles_device(c, d_rgb, width, height, static_cast<size_t>(3) * width, 0, tiles_y, quant, d_counts, d_choices,
                                         nullptr, nullptr, 0, pursuit_stream);
        if (st != MPC_OK) break;
        // Two side streams for all slots (MPC_SHARED_SIDE_STREAMS=0: one per slot): the runtime maps streams onto a handful
        // of hardware queues, and a slot stream that lands on the pursuit stream's queue lines its kernels up behind the next
        // pursuit -- with three streams in all nothing has to share.  `side_a`: stream assembly + phase 1;
        // `down`: phase 2, the container's copy, the host route's copies.
        // Measured in round 2 (all streams at one priority): per-slot streams are 4 % faster on 16 Mpixel frames (4 440 against
        // 4 270 Mpix/s) and bimodal on 2 Mpixel frames, where the chains are as long as the pursuit's tail (2 960 or 2 260 Mpix/s
        // from run to run; shared: 2 780 every time) -- so small frames share.  With the side streams prioritised (round 3) two
        // shared ones are the faster choice for large frames too (4 740 against 4 690 Mpix/s with one per slot).
        static const int shared_env = env_int("MPC_SHARED_SIDE_STREAMS", -1);
        const bool shared_sides = shared_env >= 0 ? shared_env != 0 : (c->seq_prioritised || tiles < 100000);
        hipStream_t side_a = shared_sides ? c->seq_down[0] : c->seq_down[sl];
        hipStream_t down = shared_sides ? c->seq_down[1] : c->seq_down[sl];
        hipStream_t behind = side_a;
        MPC_SEQ_TRY(hipEventRecord(c->seq_pursuit_done[sl], pursuit_stream));
        MPC_SEQ_TRY(hipStreamWaitEvent(side_a, c->seq_pursuit_done[sl], 0));
        MPC_SEQ_TRY(static_cast<hipError_t>(mpc::launch_stream_assembly(sa, behind)));
        EntropyBuffers eb = ent[sl];
        if (device_entropy) {
            eb.args.counts = d_counts;
            eb.args.symbols = sa.symbols;
            eb.args.stream_off = sa.stream_off;
            MPC_SEQ_TRY(static_cast<hipError_t>(mpc::launch_entropy_phase1(eb.args, eb.capacity_symbols, behind)));
        }
        hipEvent_t ev_comp = c->seq_events[sl][1];
        MPC_SEQ_TRY(hipEventRecord(ev_comp, behind));
        auto told = std::make_shared<std::promise<void>>();
        phase2_enqueued[sl] = told->get_future();
        const int bs = c->block_size, device = c->device;
        const uint16_t* d_symbols = sa.symbols;
        const unsigned long long* d_off = sa.stream_off;
        const size_t n_off = 6 * static_cast<size_t>(K) + 1;
        slot.frame = f;
        // the slot's worker: wait for the device, fetch the stream boundaries, then exactly the live symbols, then code them
        hipEvent_t ev_down = c->seq_events[sl][2];
        static const bool trace = env_int("MPC_TRACE", 0) != 0;
        auto now_ms = [] { return trace_ms(); };
        const double t_enq = now_ms();
        slot.result = std::async(single ? std::launch::deferred : std::launch::async, [=]() -> std::pair<uint8_t*, size_t> {
            struct Trace {
                bool on; int f; double t0, t1 = 0, t2 = 0, t3 = 0, e[3] = {0, 0, 0};
                ~Trace() {
                    if (on) std::fprintf(stderr, "[trace] frame %d enqueued %.2f | device done %.2f | symbols on host %.2f | coded %.2f | entropy: stats %.2f tables %.2f bytes %.2f\n",
                                         f, t0, t1, t2, t3, e[0], e[1], e[2]);
                }
            } tr{trace, f, t_enq};
            struct Tell {                                      // whatever happens, the enqueuing thread is released once
                std::shared_ptr<std::promise<void>> p;
                bool done = false;
                void operator()() { if (!done) p->set_value(); done = true; }
                ~Tell() { (*this)(); }
            } tell{told};
            if (hipSetDevice(device) != hipSuccess || wait_event(ev_comp, single) != hipSuccess) return {nullptr, 0};
            tr.t1 = now_ms();
            if (device_entropy) {
                uint8_t* blob = nullptr;
                size_t n = 0;
                const EntropyResult r = finish_entropy_on_device(eb, bs, width, height, K, q, down, ev_down,
                                                                 [&] { tell(); }, &blob, &n, tr.e, single);
                tr.t2 = tr.t3 = now_ms();
                if (r == EntropyResult::kDone) return {blob, n};
                if (r == EntropyResult::kFailed) return {nullptr, 0};
            }
            tell();
            if (hipMemcpyAsync(off, d_off, sizeof(unsigned long long) * n_off, hipMemcpyDeviceToHost, down) != hipSuccess) return {nullptr, 0};
            if (hipMemcpyAsync(counts, d_counts, sizeof(uint16_t) * n_tc, hipMemcpyDeviceToHost, down) != hipSuccess) return {nullptr, 0};
            if (hipEventRecord(ev_down, down) != hipSuccess || hipEventSynchronize(ev_down) != hipSuccess) return {nullptr, 0};
            const unsigned long long total = off[n_off - 1];
            if (total > 2ULL * n_tc * static_cast<unsigned long long>(K)) return {nullptr, 0};
            if (total && hipMemcpyAsync(symbols, d_symbols, sizeof(uint16_t) * total, hipMemcpyDeviceToHost, down) != hipSuccess) return {nullptr, 0};
            if (hipEventRecord(ev_down, down) != hipSuccess || hipEventSynchronize(ev_down) != hipSuccess) return {nullptr, 0};
            tr.t2 = now_ms();
            size_t n = 0;
            uint8_t* blob = mpc::encode_symbol_streams_malloc(width, height, K, bs, q, counts, symbols, off, &n);
            tr.t3 = now_ms();
            return {blob, n};
        });
    }
#undef MPC_SEQ_TRY
    for (auto& u : uploads)
        if (u.valid()) (void)u.get();
    for (int f = n_frames; f < n_frames + static_cast<int>(slots); ++f) collect(pending[f % static_cast<int>(slots)]);   // oldest first
    (void)hipStreamSynchronize(c->seq_up);
    (void)hipStreamSynchronize(pursuit_stream);
    for (size_t sl = 0; sl < slots; ++sl) (void)hipStreamSynchronize(c->seq_down[sl]);
    if (st != MPC_OK) {
        for (int f = 0; f < n_frames; ++f) { std::free(bytes[f]); bytes[f] = nullptr; nbytes[f] = 0; }
    }
    return st;
}

// ---- records that are already on the device in whole-frame order -> container (the owner of a frame in the multi-GPU path,
// after the stripe exchange), in three steps so that the caller can keep the device busy meanwhile:
//   begin    stream assembly + entropy phase 1 enqueued on `stream`; nothing is waited for
//   tables   waits for phase 1, builds the code tables, enqueues phase 2 and the container's copy on the same stream
//   collect  waits for the copy; the container
// One job per slot at a time.  mpc_records_to_container_device is the three in a row on slot 0.
namespace {
struct ContainerJob {
    int stage = 0;                      // 0 idle, 1 begun, 2 tables done
    int width = 0, height = 0;
    std::vector<double> quant;
    hipStream_t stream = nullptr;
    void* dev = nullptr;                // stream assembly buffers of this slot (grow-only)
    size_t dev_bytes = 0;
    const uint16_t* d_counts = nullptr;
    mpc::StreamArgs sa{};
    // a job's entropy-stage buffers are its own: the slots of the context belong to the frame pipeline (mpc_encode_image(s)) and
    // to mpc_code_symbol_streams_device, which may run -- and re-carve or clear their tables -- between `begin` and `collect`
    mpc_context::EntropySlot ent;
    EntropyBuffers eb;
    bool device_entropy = false;
    hipEvent_t phase1 = nullptr, done = nullptr;
    EntropyPending pending;
    uint8_t* blob = nullptr;            // the host route's result, ready at `tables`
    size_t nblob = 0;
    ~ContainerJob() {
        if (ent.dev) (void)hipFree(ent.dev);
        if (ent.host) (void)hipHostFree(ent.host);
        if (dev) (void)hipFree(dev);
        if (phase1) (void)hipEventDestroy(phase1);
        if (done) (void)hipEventDestroy(done);
        std::free(blob);
    }
};

ContainerJob* job_of(mpc_context* c, int slot) {
    if (!c->jobs[slot]) c->jobs[slot] = std::make_shared<ContainerJob>();
    return static_cast<ContainerJob*>(c->jobs[slot].get());
}

// the host route: symbols across PCIe, entropy stage on the host (synchronous)
mpc_status container_on_host(mpc_context* c, ContainerJob* j, uint8_t** bytes, size_t* nbytes) {
    const int K = c->K;
    const size_t tiles = static_cast<size_t>((j->width + 7) / 8) * ((j->height + 7) / 8), n_tc = tiles * 3;
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t counts_bytes = up(sizeof(uint16_t) * n_tc), off_bytes = up(sizeof(unsigned long long) * (6 * static_cast<size_t>(K) + 1));
    const size_t symbols_bytes = up(sizeof(uint16_t) * 2 * n_tc * K);
    const size_t host_need = counts_bytes + off_bytes + symbols_bytes;
    if (host_need > c->host_stage_bytes) {
        if (c->host_stage) (void)hipHostFree(c->host_stage);
        c->host_stage = nullptr;
        c->host_stage_bytes = 0;
        const hipError_t e = hipHostMalloc(&c->host_stage, host_need, hipHostMallocDefault);
        if (e != hipSuccess) return fail(MPC_ERR_ALLOC, "pinned staging of %zu bytes: %s", host_need, hipGetErrorString(e));
        c->host_stage_bytes = host_need;
    }
    char* hbase = static_cast<char*>(c->host_stage);
    uint16_t* counts = reinterpret_cast<uint16_t*>(hbase);
    unsigned long long* off = reinterpret_cast<unsigned long long*>(hbase + counts_bytes);
    uint16_t* symbols = reinterpret_cast<uint16_t*>(hbase + counts_bytes + off_bytes);
    hipStream_t s = j->stream;
    const size_t n_off = 6 * static_cast<size_t>(K) + 1;
    HIP_TRY(hipMemcpyAsync(off, j->sa.stream_off, sizeof(unsigned long long) * n_off, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(counts, j->d_counts, sizeof(uint16_t) * n_tc, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const unsigned long long total = off[n_off - 1];
    if (total > 2ULL * n_tc * static_cast<unsigned long long>(K)) return fail(MPC_ERR_HIP, "stream assembly returned an impossible size");
    if (total) HIP_TRY(hipMemcpyAsync(symbols, j->sa.symbols, sizeof(uint16_t) * total, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *bytes = mpc::encode_symbol_streams_malloc(j->width, j->height, K, c->block_size, j->quant.data(), counts, symbols, off, nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
}
}  // namespace

mpc_status mpc_container_job_begin(mpc_context* c, int slot, const uint16_t* d_counts, const mpc_basis_choice* d_choices, int width,
                                   int height, const double* quant, void* stream) {
    return guarded([&]() -> mpc_status {
    if (!c || !d_counts || !d_choices || slot < 0 || slot >= mpc_context::kSeqSlots) return fail(MPC_ERR_ARGUMENT, "bad argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    if (width < 1 || height < 1) return fail(MPC_ERR_ARGUMENT, "bad geometry");
    const size_t tiles = static_cast<size_t>((width + 7) / 8) * ((height + 7) / 8), n_tc = tiles * 3;
    const int K = c->K;
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    HIP_TRY(hipSetDevice(c->device));
    ContainerJob* j = job_of(c, slot);
    if (j->stage != 0) return fail(MPC_ERR_ARGUMENT, "container job slot %d is busy", slot);
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    const size_t off_bytes = up(sizeof(unsigned long long) * (6 * static_cast<size_t>(K) + 1));
    const size_t symbols_bytes = up(sizeof(uint16_t) * 2 * n_tc * K);
    const size_t live_bytes = up(sizeof(unsigned) * mpc::stream_workspace_words(static_cast<long long>(tiles), K));
    const size_t sizes_bytes = up(sizeof(unsigned) * 3 * K), dc_bytes = up(sizeof(uint16_t) * n_tc);
    const size_t dev_need = live_bytes + sizes_bytes + off_bytes + symbols_bytes + dc_bytes;
    if (dev_need > j->dev_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        if (j->dev) (void)hipFree(j->dev);
        j->dev = nullptr;
        j->dev_bytes = 0;
        const hipError_t e = hipMalloc(&j->dev, dev_need);
        if (e != hipSuccess) return fail(MPC_ERR_ALLOC, "device staging of %zu bytes: %s", dev_need, hipGetErrorString(e));
        j->dev_bytes = dev_need;
    }
    if (!j->phase1) {
        HIP_TRY(hipEventCreateWithFlags(&j->phase1, hipEventDisableTiming | hipEventBlockingSync));
        HIP_TRY(hipEventCreateWithFlags(&j->done, hipEventDisableTiming | hipEventBlockingSync));
    }
    char* dbase = static_cast<char*>(j->dev);
    mpc::StreamArgs& sa = j->sa;
    sa = mpc::StreamArgs{};
    sa.counts = d_counts;
    sa.choices = reinterpret_cast<const uint32_t*>(d_choices);
    sa.tiles = static_cast<long long>(tiles);
    sa.K = K;
    sa.block_live = reinterpret_cast<unsigned*>(dbase);
    sa.sizes = reinterpret_cast<unsigned*>(dbase + live_bytes);
    sa.stream_off = reinterpret_cast<unsigned long long*>(dbase + live_bytes + sizes_bytes);
    sa.symbols = reinterpret_cast<uint16_t*>(dbase + live_bytes + sizes_bytes + off_bytes);
    sa.dc_tmp = reinterpret_cast<uint16_t*>(dbase + live_bytes + sizes_bytes + off_bytes + symbols_bytes);
    j->width = width;
    j->height = height;
    j->d_counts = d_counts;
    j->stream = static_cast<hipStream_t>(stream);
    const double* q = quant ? quant : c->quant.data();
    j->quant.assign(q, q + 3 * static_cast<size_t>(K));
    j->device_entropy = !host_entropy_forced();
    if (j->device_entropy) {
        const mpc_status es = entropy_buffers(j->ent, tiles, K, &j->eb);
        if (es != MPC_OK) return es;
    }
    const int err = mpc::launch_stream_assembly(sa, j->stream);
    if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    if (j->device_entropy) {
        j->eb.args.counts = d_counts;
        j->eb.args.symbols = sa.symbols;
        j->eb.args.stream_off = sa.stream_off;
        const int e1 = mpc::launch_entropy_phase1(j->eb.args, j->eb.capacity_symbols, j->stream);
        if (e1 != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e1)));
    }
    HIP_TRY(hipEventRecord(j->phase1, j->stream));
    j->stage = 1;
    return MPC_OK;
    });
}

mpc_status mpc_container_job_tables(mpc_context* c, int slot) {
    return guarded([&]() -> mpc_status {
    if (!c || slot < 0 || slot >= mpc_context::kSeqSlots) return fail(MPC_ERR_ARGUMENT, "bad argument");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    ContainerJob* j = job_of(c, slot);
    if (j->stage != 1) return fail(MPC_ERR_ARGUMENT, "container job slot %d has not begun", slot);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(j->phase1));
    std::free(j->blob);
    j->blob = nullptr;
    j->nblob = 0;
    EntropyResult r = EntropyResult::kNeedsHost;
    if (j->device_entropy) {
        r = entropy_tables(j->eb, c->block_size, j->width, j->height, c->K, j->quant.data(), j->stream, j->done, &j->pending);
        if (r == EntropyResult::kFailed) { j->stage = 0; return fail(MPC_ERR_HIP, "device entropy stage failed: %s", hipGetErrorString(hipGetLastError())); }
    }
    if (r == EntropyResult::kNeedsHost) {
        j->device_entropy = false;
        const mpc_status st = container_on_host(c, j, &j->blob, &j->nblob);
        if (st != MPC_OK) { j->stage = 0; return st; }
    }
    j->stage = 2;
    return MPC_OK;
    });
}

mpc_status mpc_container_job_collect(mpc_context* c, int slot, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!c || !bytes || !nbytes || slot < 0 || slot >= mpc_context::kSeqSlots) return fail(MPC_ERR_ARGUMENT, "bad argument");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    ContainerJob* j = job_of(c, slot);
    if (j->stage != 2) return fail(MPC_ERR_ARGUMENT, "container job slot %d has no tables yet", slot);
    HIP_TRY(hipSetDevice(c->device));
    j->stage = 0;
    if (!j->device_entropy) {
        *bytes = j->blob;
        *nbytes = j->nblob;
        j->blob = nullptr;
        j->nblob = 0;
        return MPC_OK;
    }
    const EntropyResult r = entropy_collect(j->eb, j->pending, j->done, bytes, nbytes);
    return r == EntropyResult::kDone ? MPC_OK : fail(MPC_ERR_HIP, "device entropy stage failed: %s", hipGetErrorString(hipGetLastError()));
    });
}

mpc_status mpc_container_job_cancel(mpc_context* c, int slot) {
    return guarded([&]() -> mpc_status {
    if (!c || slot < 0 || slot >= mpc_context::kSeqSlots) return fail(MPC_ERR_ARGUMENT, "bad argument");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    if (!c->jobs[slot]) return MPC_OK;
    ContainerJob* j = job_of(c, slot);
    if (j->stage != 0 && c->device >= 0) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(j->stream));          // whatever the job has enqueued has left its buffers
    }
    std::free(j->blob);
    j->blob = nullptr;
    j->nblob = 0;
    j->stage = 0;
    return MPC_OK;
    });
}

mpc_status mpc_interleave_stripe_device(mpc_context* c, const uint16_t* d_part_counts, const mpc_basis_choice* d_part_choices, int width,
                                        int height, int tile_row_begin, int tile_row_end, uint16_t* d_frame_counts,
                                        mpc_basis_choice* d_frame_choices, void* stream) {
    if (!c || !d_part_counts || !d_part_choices || !d_frame_counts || !d_frame_choices) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    if (width < 1 || height < 1) return fail(MPC_ERR_ARGUMENT, "bad geometry");
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    if (tile_row_begin < 0 || tile_row_end > tiles_y || tile_row_begin >= tile_row_end)
        return fail(MPC_ERR_ARGUMENT, "tile rows [%d,%d) outside 0..%d", tile_row_begin, tile_row_end, tiles_y);
    HIP_TRY(hipSetDevice(c->device));
    const int err = mpc::launch_interleave_stripe(d_part_counts, reinterpret_cast<const uint32_t*>(d_part_choices), tiles_x, tiles_y, tile_row_begin,
                                                  tile_row_end - tile_row_begin, c->K, d_frame_counts, reinterpret_cast<uint32_t*>(d_frame_choices), stream);
    if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    return MPC_OK;
}

mpc_status mpc_records_to_container_device(mpc_context* c, const uint16_t* d_counts, const mpc_basis_choice* d_choices, int width,
                                           int height, const double* quant, void* stream, uint8_t** bytes, size_t* nbytes) {
    if (!bytes || !nbytes) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (!c) return fail(MPC_ERR_ARGUMENT, "null context");
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    mpc_status st = mpc_container_job_begin(c, 0, d_counts, d_choices, width, height, quant, stream);
    if (st == MPC_OK) st = mpc_container_job_tables(c, 0);
    if (st == MPC_OK) st = mpc_container_job_collect(c, 0, bytes, nbytes);
    return st;
}

// The entropy stage alone, on streams the caller holds in host memory (what mpc_assemble_symbol_streams codes on the host):
// upload, device entropy stage, container bytes.  *route (optional): 0 = coded on the device, 1 = the host route was taken.
mpc_status mpc_code_symbol_streams_device(mpc_context* c, int width, int height, const double* quant, const uint16_t* counts,
                                          const uint16_t* symbols, const unsigned long long* stream_off, uint8_t** bytes, size_t* nbytes,
                                          int* route) {
    return guarded([&]() -> mpc_status {
    if (!c || !counts || !stream_off || !bytes || !nbytes || width < 1 || height < 1) return fail(MPC_ERR_ARGUMENT, "bad argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    const int K = c->K;
    const size_t tiles = static_cast<size_t>((width + 7) / 8) * ((height + 7) / 8), n_tc = tiles * 3;
    for (int s = 0; s < 6 * K; ++s)
        if (stream_off[s + 1] < stream_off[s]) return fail(MPC_ERR_ARGUMENT, "stream offsets must not decrease");
    const unsigned long long total = stream_off[6 * K];
    if (stream_off[0] != 0 || total > 2ULL * n_tc * K || (total && !symbols)) return fail(MPC_ERR_ARGUMENT, "streams larger than a frame of this size can hold");
    const double* q = quant ? quant : c->quant.data();
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);
    HIP_TRY(hipSetDevice(c->device));
    if (route) *route = 1;
    if (!host_entropy_forced()) {
        EntropyBuffers eb;
        const mpc_status es = entropy_buffers(c->ent[0], tiles, K, &eb);
        if (es != MPC_OK) return es;
        struct Temp {
            void* p = nullptr;
            ~Temp() { if (p) (void)hipFree(p); }
        } d_counts, d_symbols, d_off;
        const size_t n_off = 6 * static_cast<size_t>(K) + 1;
        HIP_TRY(hipMalloc(&d_counts.p, sizeof(uint16_t) * n_tc));
        HIP_TRY(hipMalloc(&d_symbols.p, sizeof(uint16_t) * (total ? total : 1)));
        HIP_TRY(hipMalloc(&d_off.p, sizeof(unsigned long long) * n_off));
        HIP_TRY(hipMemcpy(d_counts.p, counts, sizeof(uint16_t) * n_tc, hipMemcpyHostToDevice));
        if (total) HIP_TRY(hipMemcpy(d_symbols.p, symbols, sizeof(uint16_t) * total, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_off.p, stream_off, sizeof(unsigned long long) * n_off, hipMemcpyHostToDevice));
        eb.args.counts = static_cast<const uint16_t*>(d_counts.p);
        eb.args.symbols = static_cast<const uint16_t*>(d_symbols.p);
        eb.args.stream_off = static_cast<const unsigned long long*>(d_off.p);
        const int e1 = mpc::launch_entropy_phase1(eb.args, eb.capacity_symbols, nullptr);
        if (e1 != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(e1)));
        HIP_TRY(hipStreamSynchronize(nullptr));
        hipEvent_t done = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        const EntropyResult r = finish_entropy_on_device(eb, c->block_size, width, height, K, q, nullptr, done, nullptr, bytes, nbytes);
        (void)hipEventDestroy(done);
        HIP_TRY(hipDeviceSynchronize());
        if (r == EntropyResult::kDone) {
            if (route) *route = 0;
            return MPC_OK;
        }
        if (r == EntropyResult::kFailed) return fail(MPC_ERR_HIP, "device entropy stage failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *bytes = mpc::encode_symbol_streams_malloc(width, height, K, c->block_size, q, counts, symbols, stream_off, nbytes);
    return *bytes ? MPC_OK : fail(MPC_ERR_ALLOC, "out of memory");
    });
}

mpc_status mpc_encode_images(mpc_context* c, const uint8_t* const* rgb_frames, int n_frames, int width, int height,
                             const double* quant, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status { return encode_sequence(c, rgb_frames, false, n_frames, width, height, quant, bytes, nbytes); });
}

mpc_status mpc_encode_images_device(mpc_context* c, const uint8_t* const* d_rgb_frames, int n_frames, int width, int height,
                                    const double* quant, uint8_t** bytes, size_t* nbytes) {
    return guarded([&]() -> mpc_status { return encode_sequence(c, d_rgb_frames, true, n_frames, width, height, quant, bytes, nbytes); });
}

// compressed::encodeImage: one frame through the same stages
mpc_status mpc_encode_image(mpc_context* c, const uint8_t* rgb, int width, int height, const double* quant,
                            uint8_t** bytes, size_t* nbytes) {
    if (!rgb || !bytes || !nbytes) return fail(MPC_ERR_ARGUMENT, "null argument");
    return guarded([&]() -> mpc_status { return encode_sequence(c, &rgb, false, 1, width, height, quant, bytes, nbytes); });
}

mpc_status mpc_encode_image_device(mpc_context* c, const uint8_t* d_rgb, int width, int height, const double* quant,
                                   uint8_t** bytes, size_t* nbytes) {
    if (!d_rgb || !bytes || !nbytes) return fail(MPC_ERR_ARGUMENT, "null argument");
    return guarded([&]() -> mpc_status { return encode_sequence(c, &d_rgb, true, 1, width, height, quant, bytes, nbytes); });
}

// FromCoeffsDynamic + RGBFromYUV for whole tiles on the device; d_quant: [3][K] doubles on the device
static mpc_status decode_tiles_on_device(mpc_context* c, const uint16_t* d_counts, const uint32_t* d_choices, const double* d_quant,
                                         int K, int width, int height, uint8_t* d_rgb, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!c->d_flag) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->d_flag), sizeof(int)));
    HIP_TRY(hipMemsetAsync(c->d_flag, 0, sizeof(int), s));
    mpc::DecodeParams p{};
    p.counts = d_counts;
    p.choices = d_choices;
    p.quant = d_quant;
    p.K = K;
    p.width = width;
    p.height = height;
    p.tiles_x = (width + 7) / 8;
    p.tiles_y = (height + 7) / 8;
    p.rgb = d_rgb;
    p.error_flag = c->d_flag;
    p.fast = c->fast ? 1 : 0;
    const int err = mpc::launch_decode(dict_device(c), p, stream);
    if (err != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(err)));
    return MPC_OK;
}

mpc_status mpc_decode_tiles_device(mpc_context* c, const uint16_t* d_counts, const mpc_basis_choice* d_choices,
                                   const double* quant, int width, int height, uint8_t* d_rgb, void* stream) {
    if (!c || !d_counts || !d_choices || !d_rgb) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device");
    if (width < 1 || height < 1) return fail(MPC_ERR_ARGUMENT, "bad geometry");
    HIP_TRY(hipSetDevice(c->device));
    const double* d_q = nullptr;
    if (const mpc_status qs = call_quant(c, quant, static_cast<hipStream_t>(stream), &d_q); qs != MPC_OK) return qs;
    return decode_tiles_on_device(c, d_counts, reinterpret_cast<const uint32_t*>(d_choices), d_q, c->K, width, height, d_rgb,
                                  stream);
}

// compressed::decodeImage: container parsing on the host, tile reconstruction on the device.  The stream's own
// K and quantisation table are used (they need not match the context's); there is no host reconstruction.
mpc_status mpc_decode_image(const mpc_context* cc, const uint8_t* bytes, size_t nbytes, uint8_t** rgb, int* width, int* height) {
    return guarded([&]() -> mpc_status {
    if (!cc || !bytes || !rgb || !width || !height) return fail(MPC_ERR_ARGUMENT, "null argument");
    mpc_context* c = const_cast<mpc_context*>(cc);
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    static const bool trace = env_int("MPC_TRACE", 0) != 0;
    const double t_begin = trace_ms();
    mpc::Streams s;
    if (!mpc::read_compressed(bytes, nbytes, s)) return fail(MPC_ERR_BITSTREAM, "Invalid input data");
    const double t_parsed = trace_ms();
    std::lock_guard<std::recursive_mutex> one_host_call(c->host_calls);     // concurrent decodes share the staging buffers
    if (s.block_size != c->block_size) return fail(MPC_ERR_ARGUMENT, "stream block size %d, context block size %d", s.block_size, c->block_size);
    // The streams -- not the records -- cross PCIe (21 MB instead of 97 for a 16 Mpixel K = 32 frame) through the context's pinned
    // host buffer and device staging area (grow-only, shared with the encoder); the records are rebuilt on the device
    // (mp_stream_gather_kernel: the stream assembly's positions, read the other way).
    HIP_TRY(hipSetDevice(c->device));
    const size_t n_tc = s.lengths.size();
    const size_t tiles = n_tc / 3;
    if (n_tc != 3 * tiles || s.codes.size() != static_cast<size_t>(6 * s.K)) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
    auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
    std::vector<size_t> stream_at(static_cast<size_t>(6 * s.K) + 1, 0);
    for (int i = 0; i < 6 * s.K; ++i) {
        if ((i & 1) && s.codes[i].size() != s.codes[i - 1].size()) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
        stream_at[i + 1] = stream_at[i] + s.codes[i].size();
    }
    const size_t n_symbols = stream_at[6 * s.K];
    for (size_t o = 0; o < n_tc; ++o)
        if (s.lengths[o] > s.K) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
    const size_t counts_bytes = up(sizeof(uint16_t) * n_tc), symbols_bytes = up(sizeof(uint16_t) * (n_symbols ? n_symbols : 1));
    const size_t quant_bytes = up(sizeof(double) * 3 * s.K);
    const size_t upload_bytes = counts_bytes + symbols_bytes + quant_bytes;
    const size_t choices_bytes = up(sizeof(uint32_t) * n_tc * s.K);
    const size_t live_bytes = up(sizeof(unsigned) * mpc::stream_workspace_words(static_cast<long long>(tiles), s.K));
    const size_t sizes_bytes = up(sizeof(unsigned) * 3 * s.K);
    const size_t px = static_cast<size_t>(s.width) * s.height * 3;
    const size_t dev_bytes = upload_bytes + choices_bytes + live_bytes + sizes_bytes + up(px);
    const size_t pinned_bytes = std::max(upload_bytes, up(px));      // the streams on their way in, later the pixels on their way out
    if (pinned_bytes > c->host_stage_bytes) {
        if (c->host_stage) (void)hipHostFree(c->host_stage);
        c->host_stage = nullptr;
        c->host_stage_bytes = 0;
        const hipError_t ea = hipHostMalloc(&c->host_stage, pinned_bytes, hipHostMallocDefault);
        if (ea != hipSuccess) return fail(MPC_ERR_ALLOC, "pinned staging: %s", hipGetErrorString(ea));
        c->host_stage_bytes = pinned_bytes;
    }
    if (dev_bytes > c->stage_bytes) {
        if (c->stage) (void)hipFree(c->stage);
        c->stage = nullptr;
        c->stage_bytes = 0;
        const hipError_t ea = hipMalloc(&c->stage, dev_bytes);
        if (ea != hipSuccess) return fail(MPC_ERR_ALLOC, "device staging: %s", hipGetErrorString(ea));
        c->stage_bytes = dev_bytes;
    }
    char* hbase = static_cast<char*>(c->host_stage);
    uint16_t* counts = reinterpret_cast<uint16_t*>(hbase);
    uint16_t* symbols = reinterpret_cast<uint16_t*>(hbase + counts_bytes);
    double* q = reinterpret_cast<double*>(hbase + counts_bytes + symbols_bytes);
    mpc::parallel_jobs(6 * s.K + 1, [&](int job) {
        if (job == 0) std::memcpy(counts, s.lengths.data(), sizeof(uint16_t) * n_tc);
        else if (!s.codes[job - 1].empty())
            std::memcpy(symbols + stream_at[job - 1], s.codes[job - 1].data(), sizeof(uint16_t) * s.codes[job - 1].size());
    });
    for (int ch = 0; ch < 3; ++ch)
        for (int i = 0; i < s.K; ++i) q[ch * s.K + i] = static_cast<double>(s.quant[ch][i]);
    char* dbase = static_cast<char*>(c->stage);
    uint16_t* d_counts = reinterpret_cast<uint16_t*>(dbase);
    uint16_t* d_symbols = reinterpret_cast<uint16_t*>(dbase + counts_bytes);
    double* d_q = reinterpret_cast<double*>(dbase + counts_bytes + symbols_bytes);
    uint32_t* d_choices = reinterpret_cast<uint32_t*>(dbase + upload_bytes);
    uint8_t* d_rgb = reinterpret_cast<uint8_t*>(dbase + upload_bytes + choices_bytes + live_bytes + sizes_bytes);
    const double t_staged = trace_ms();
    HIP_TRY(hipMemcpyAsync(dbase, hbase, upload_bytes, hipMemcpyHostToDevice, nullptr));
    mpc::StreamArgs sa{};
    sa.counts = d_counts;
    sa.tiles = static_cast<long long>(tiles);
    sa.K = s.K;
    sa.block_live = reinterpret_cast<unsigned*>(dbase + upload_bytes + choices_bytes);
    sa.sizes = reinterpret_cast<unsigned*>(dbase + upload_bytes + choices_bytes + live_bytes);
    sa.symbols = d_symbols;
    const int ge = mpc::launch_stream_gather(sa, d_choices, nullptr);
    if (ge != 0) return fail(MPC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(static_cast<hipError_t>(ge)));
    const mpc_status st = decode_tiles_on_device(c, d_counts, d_choices, d_q, s.K, s.width, s.height, d_rgb, nullptr);
    if (st != MPC_OK) return st;
    HIP_TRY(hipDeviceSynchronize());
    const double t_device = trace_ms();
    int flag = 0;
    HIP_TRY(hipMemcpy(&flag, c->d_flag, sizeof(int), hipMemcpyDeviceToHost));
    if (flag) return fail(MPC_ERR_BITSTREAM, "Invalid bitstream");
    uint8_t* out = static_cast<uint8_t*>(std::malloc(px ? px : 1));
    if (!out) return fail(MPC_ERR_ALLOC, "out of memory");
    // through pinned memory (a copy into fresh pageable pages is staged by the runtime on one thread), then a few threads fault
    // the caller's pages in and copy
    const hipError_t e = hipMemcpy(c->host_stage, d_rgb, px, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { std::free(out); return fail(MPC_ERR_HIP, "HIP failure: %s", hipGetErrorString(e)); }
    {
        const uint8_t* src = static_cast<const uint8_t*>(c->host_stage);
        const size_t piece = ((px + 15) / 16 + 4095) & ~static_cast<size_t>(4095);
        mpc::parallel_jobs(piece ? static_cast<int>((px + piece - 1) / piece) : 0, [&](int k) {
            const size_t lo = piece * static_cast<size_t>(k), hi = std::min(px, lo + piece);
            std::memcpy(out + lo, src + lo, hi - lo);
        });
    }
    if (trace)
        std::fprintf(stderr, "[trace] decode: parse %.2f ms | streams staged %.2f | upload + gather + reconstruct %.2f | pixels to the caller %.2f\n",
                     t_parsed - t_begin, t_staged - t_parsed, t_device - t_staged, trace_ms() - t_device);
    *rgb = out;
    *width = s.width;
    *height = s.height;
    return MPC_OK;
    });
}

// ---- "-s" patch statistics (Compression.cpp:200-302) ----
struct mpc_patch_stats {
    mpc_context* ctx;
    mpc::PatchStats stats;
    std::vector<uint8_t> mosaic;
    std::vector<uint16_t> counts;
    std::vector<uint32_t> choices;
    mpc_patch_stats(mpc_context* c, unsigned seed) : ctx(c), stats(c->K, c->block_size, seed) {}
};

mpc_status mpc_patch_stats_create(mpc_context* c, unsigned seed, mpc_patch_stats** out) {
    if (!c || !out) return fail(MPC_ERR_ARGUMENT, "null argument");
    if (c->device < 0) return fail(MPC_ERR_NO_DEVICE, "context was created without a device; there is no CPU fallback");
    try {
        *out = new mpc_patch_stats(c, seed);
    } catch (const std::bad_alloc&) { return fail(MPC_ERR_ALLOC, "out of memory"); }
    return MPC_OK;
}

void mpc_patch_stats_destroy(mpc_patch_stats* s) { delete s; }

mpc_status mpc_patch_stats_add_image(mpc_patch_stats* s, const uint8_t* rgb, int width, int height, int patches) {
    return guarded([&]() -> mpc_status {
    if (!s || !rgb) return fail(MPC_ERR_ARGUMENT, "null argument");
    const int bs = s->stats.block_size, K = s->stats.K;
    if (width < bs || height < bs || patches <= 0) return MPC_OK;            // Compression.cpp:233-236
    if (width == bs || height == bs) return fail(MPC_ERR_ARGUMENT, "image of exactly one block: rand() %% 0 in the reference");
    std::vector<int> xs, ys;
    s->stats.sample_origins(width, height, patches, xs, ys);
    // the patches as the tiles of a one-tile-high mosaic: tile p = patch p (tile order tx*1 + 0)
    const size_t row = static_cast<size_t>(patches) * bs * 3;
    s->mosaic.resize(row * bs);
    for (int p = 0; p < patches; ++p)
        for (int dy = 0; dy < bs; ++dy)
            std::memcpy(s->mosaic.data() + dy * row + static_cast<size_t>(p) * bs * 3,
                        rgb + 3 * (static_cast<size_t>(ys[p] + dy) * width + xs[p]), static_cast<size_t>(bs) * 3);
    s->counts.resize(static_cast<size_t>(patches) * 3);
    s->choices.resize(static_cast<size_t>(patches) * 3 * K);
    const std::vector<double> ones(3 * static_cast<size_t>(K), 1.0);          // Compression.cpp:221-225
    const mpc_status st = mpc_encode_tiles(s->ctx, s->mosaic.data(), patches * bs, bs, row, 0, 1, ones.data(), s->counts.data(),
                                           reinterpret_cast<mpc_basis_choice*>(s->choices.data()), nullptr, nullptr);
    if (st != MPC_OK) return st;
    s->stats.accumulate(s->counts.data(), s->choices.data(), patches);
    return MPC_OK;
    });
}

mpc_status mpc_patch_stats_read(const mpc_patch_stats* s, double* out) {
    if (!s || !out) return fail(MPC_ERR_ARGUMENT, "null argument");
    const int K = s->stats.K;
    for (int ch = 0; ch < 3; ++ch)
        for (int kind = 0; kind < 2; ++kind)
            for (int i = 0; i < K; ++i) {
                const mpc::RunningStat& r = kind == 0 ? s->stats.coeff[ch][i] : s->stats.select[ch][i];
                double* o = out + ((static_cast<size_t>(ch) * 2 + kind) * K + i) * 5;
                o[0] = r.N; o[1] = r.min; o[2] = r.max; o[3] = r.mean; o[4] = r.sumSq;
            }
    return MPC_OK;
}

mpc_status mpc_patch_stats_report(const mpc_patch_stats* s, char** text, size_t* nbytes) {
    return guarded([&]() -> mpc_status {
    if (!s || !text || !nbytes) return fail(MPC_ERR_ARGUMENT, "null argument");
    const std::string r = s->stats.report();
    char* p = static_cast<char*>(std::malloc(r.size() + 1));
    if (!p) return fail(MPC_ERR_ALLOC, "out of memory");
    std::memcpy(p, r.c_str(), r.size() + 1);
    *text = p;
    *nbytes = r.size();
    return MPC_OK;
    });
}

int mpc_format_double(double v, char* buf, int cap) {
    const std::string t = mpc::format_double(v);
    if (!buf || cap <= static_cast<int>(t.size())) return -1;
    std::memcpy(buf, t.c_str(), t.size() + 1);
    return static_cast<int>(t.size());
}

double mpc_psnr(const uint8_t* original, const uint8_t* decoded, int width, int height) {
    return mpc::psnr(original, decoded, width, height);
}

}  // extern "C"
