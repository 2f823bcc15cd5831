// mpcodec_multi.cpp -- product: compressed::encodeImage for a sequence of frames on SEVERAL GPUs of one node from ONE process
// (mpc_encode_images_multi, include/mpcodec.h).  A client of the single-device C ABI and of the HIP runtime only.
//
// The north_star's partition: every frame's tile rows are striped over the devices (device r encodes tile rows [begin_r, end_r) of
// every frame of a step, one launch: mpc_encode_batch_device).  A container is one bit stream per frame in the reference's
// x-outer / y-inner tile order (CompressedImage.cpp:535-537), with DC chains and run lengths crossing the stripe boundaries
// (:428-453), so a frame's records must meet on one device: with n devices a step takes n frames, frame f of the step is OWNED
// by device f, the owner pulls the other devices' stripes of its frame (hipMemcpyPeerAsync: xGMI between the GPUs of a node),
// puts them into frame order (mpc_interleave_stripe_device) and runs stream assembly + entropy stage for it (mpc_container_job_*).
// Every device does an equal share of every stage; no collective, no host copy of records.
//
// One host thread per device ("lane") enqueues that device's work on the lane's stream and builds its frame's code tables;
// consecutive steps are software-pipelined per lane like sharding.StripedEncoder.run (step s: upload + stripe encode + pulls +
// stream assembly + entropy phase 1 enqueued; then step s-1's tables built and phase 2 enqueued; then step s-2's container
// collected).  Between lanes: HIP events for the device-side order (a pull waits for the source's encode, an encode waits until
// the stripes it overwrites have been pulled) and two arrays of step counters for the host-side order (an event may only be waited
// for once it has been recorded).
#include "../../include/mpcodec.h"
#include "mpc_internal.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <future>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Lane {
    mpc_context* ctx = nullptr;
    int device = -1;
    int begin = 0, end = 0;                       // this lane's tile rows
    hipStream_t stream = nullptr, up_stream = nullptr;     // the tile encodes; the uploads
    hipStream_t side = nullptr;                          // everything behind a tile encode: pulls, interleave, the container job
    uint8_t* d_rgb[2] = {nullptr, nullptr};       // [frames of a step][H][W][3]; only this lane's rows are ever written; by step parity
    uint8_t* h_rgb[2] = {nullptr, nullptr};       // pinned: this lane's rows of the frames of a step
    uint16_t* d_counts[2] = {nullptr, nullptr};   // this lane's stripes of a step's frames [frame][tiles_x * rows][3]; by step parity
    mpc_basis_choice* d_choices[2] = {nullptr, nullptr};
    std::vector<uint16_t*> d_part_counts;         // as an owner: lane q's stripe of my frame, pulled here
    std::vector<mpc_basis_choice*> d_part_choices;
    uint16_t* d_frame_counts[3] = {nullptr, nullptr, nullptr};     // whole-frame records, one per container job slot
    mpc_basis_choice* d_frame_choices[3] = {nullptr, nullptr, nullptr};
    hipEvent_t uploaded[2] = {nullptr, nullptr}, encoded[2] = {nullptr, nullptr}, pulled[2] = {nullptr, nullptr};
    std::atomic<long> encoded_step{-1}, pulled_step{-1};           // last step whose event has been RECORDED
    std::string error;
};

void stripe_bounds(int tiles_y, int n, int r, int* begin, int* end) {      // remainder to the first lanes (540 rows over 8: 68 x 4 + 67 x 4)
    const int base = tiles_y / n, rem = tiles_y % n;
    *begin = r * base + std::min(r, rem);
    *end = *begin + base + (r < rem ? 1 : 0);
}

std::string text(const char* fmt, ...) {
    char buf[400];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}

void release(Lane& l) {
    if (l.device < 0) return;
    (void)hipSetDevice(l.device);
    if (l.stream) (void)hipStreamSynchronize(l.stream);
    if (l.up_stream) (void)hipStreamSynchronize(l.up_stream);
    if (l.side) (void)hipStreamSynchronize(l.side);
    for (int p = 0; p < 2; ++p) {
        (void)hipFree(l.d_rgb[p]); (void)hipFree(l.d_counts[p]); (void)hipFree(l.d_choices[p]);
        if (l.h_rgb[p]) (void)hipHostFree(l.h_rgb[p]);
        if (l.uploaded[p]) (void)hipEventDestroy(l.uploaded[p]);
        if (l.encoded[p]) (void)hipEventDestroy(l.encoded[p]);
        if (l.pulled[p]) (void)hipEventDestroy(l.pulled[p]);
    }
    for (auto* p : l.d_part_counts) (void)hipFree(p);
    for (auto* p : l.d_part_choices) (void)hipFree(p);
    for (int j = 0; j < 3; ++j) { (void)hipFree(l.d_frame_counts[j]); (void)hipFree(l.d_frame_choices[j]); }
    if (l.stream) (void)hipStreamDestroy(l.stream);
    if (l.up_stream) (void)hipStreamDestroy(l.up_stream);
    if (l.side) (void)hipStreamDestroy(l.side);
}

}  // namespace

extern "C" mpc_status mpc_encode_images_multi(mpc_context* const* ctxs, int n_devices, const uint8_t* const* rgb_frames, int n_frames,
                                              int width, int height, const double* quant, uint8_t** bytes, size_t* nbytes) {
    if (!ctxs || !rgb_frames || !bytes || !nbytes || n_devices < 1 || n_devices > 64 || n_frames < 1 || width < 1 || height < 1)
        return MPC_ERR_ARGUMENT;
    for (int f = 0; f < n_frames; ++f) {
        if (!rgb_frames[f]) return MPC_ERR_ARGUMENT;
        bytes[f] = nullptr;
        nbytes[f] = 0;
    }
    const int tiles_x = (width + 7) / 8, tiles_y = (height + 7) / 8;
    if (n_devices == 1) return mpc_encode_images(ctxs[0], rgb_frames, n_frames, width, height, quant, bytes, nbytes);
    if (n_devices > tiles_y) return MPC_ERR_ARGUMENT;                    // a lane without a tile row
    const int N = n_devices, K = mpc_context_K(ctxs[0]);
    for (int r = 0; r < N; ++r) {
        if (!ctxs[r] || mpc_context_device(ctxs[r]) < 0) return MPC_ERR_NO_DEVICE;
        if (mpc_context_K(ctxs[r]) != K || mpc_context_block_size(ctxs[r]) != 8) return MPC_ERR_ARGUMENT;
        for (int q = 0; q < r; ++q)
            if (ctxs[q] == ctxs[r]) return MPC_ERR_ARGUMENT;             // one context per lane (two lanes may share a DEVICE)
    }
    const size_t row_bytes = static_cast<size_t>(3) * width, frame_bytes = row_bytes * height;
    const size_t tiles = static_cast<size_t>(tiles_x) * tiles_y;
    std::vector<Lane> lanes(static_cast<size_t>(N));
    std::atomic<bool> failed{false};
    const long steps = (n_frames + N - 1) / N;

    auto setup = [&](int r) -> bool {
        Lane& l = lanes[static_cast<size_t>(r)];
        l.ctx = ctxs[r];
        l.device = mpc_context_device(ctxs[r]);
        stripe_bounds(tiles_y, N, r, &l.begin, &l.end);
        const size_t stripe_tiles = static_cast<size_t>(tiles_x) * (l.end - l.begin);
        const int y0 = l.begin * 8, y1 = std::min(height, l.end * 8);
        bool ok = hipSetDevice(l.device) == hipSuccess && hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&l.up_stream, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&l.side, hipStreamNonBlocking) == hipSuccess;
        for (int p = 0; p < 2 && ok; ++p) {
            ok = hipMalloc(reinterpret_cast<void**>(&l.d_rgb[p]), frame_bytes * N) == hipSuccess &&
                 hipHostMalloc(reinterpret_cast<void**>(&l.h_rgb[p]), row_bytes * (y1 - y0) * N, hipHostMallocDefault) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&l.d_counts[p]), sizeof(uint16_t) * 3 * stripe_tiles * N) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&l.d_choices[p]), sizeof(mpc_basis_choice) * 3 * K * stripe_tiles * N) == hipSuccess &&
                 hipEventCreateWithFlags(&l.uploaded[p], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&l.encoded[p], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&l.pulled[p], hipEventDisableTiming) == hipSuccess;
        }
        l.d_part_counts.assign(static_cast<size_t>(N), nullptr);
        l.d_part_choices.assign(static_cast<size_t>(N), nullptr);
        for (int q = 0; q < N && ok; ++q) {
            if (q == r) continue;                                        // my own stripe is interleaved from where the encode left it
            int b, e;
            stripe_bounds(tiles_y, N, q, &b, &e);
            const size_t part_tiles = static_cast<size_t>(tiles_x) * (e - b);
            ok = hipMalloc(reinterpret_cast<void**>(&l.d_part_counts[static_cast<size_t>(q)]), sizeof(uint16_t) * 3 * part_tiles) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&l.d_part_choices[static_cast<size_t>(q)]), sizeof(mpc_basis_choice) * 3 * K * part_tiles) == hipSuccess;
            // direct access to the source lane's device, where the platform offers it (otherwise the runtime stages the copy)
            const int other = mpc_context_device(ctxs[q]);
            if (ok && other != l.device) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, l.device, other) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(other, 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) ok = false;
                    (void)hipGetLastError();
                }
            }
        }
        for (int j = 0; j < 3 && ok; ++j)
            ok = hipMalloc(reinterpret_cast<void**>(&l.d_frame_counts[j]), sizeof(uint16_t) * 3 * tiles) == hipSuccess &&
                 hipMalloc(reinterpret_cast<void**>(&l.d_frame_choices[j]), sizeof(mpc_basis_choice) * 3 * K * tiles) == hipSuccess;
        if (!ok) l.error = text("lane %d (device %d): set-up failed: %s", r, l.device, hipGetErrorString(hipGetLastError()));
        return ok;
    };
    for (int r = 0; r < N; ++r)
        if (!setup(r)) failed = true;

    // host-side order between lanes: wait until `counter` has reached `step` (or somebody failed)
    auto await = [&](const std::atomic<long>& counter, long step) {
        while (counter.load(std::memory_order_acquire) < step && !failed.load(std::memory_order_relaxed)) std::this_thread::yield();
        return !failed.load(std::memory_order_relaxed);
    };

    auto run_lane = [&](int r) {
        Lane& me = lanes[static_cast<size_t>(r)];
        auto fail = [&](const std::string& why) {
            if (me.error.empty()) me.error = why;
            failed = true;
        };
#define LANE_HIP(call)                                                                                        \
    {                                                                                                         \
        const hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) { fail(text("lane %d: %s failed: %s", r, #call, hipGetErrorString(e_))); break; } \
    }
#define LANE_MPC(call)                                                                                  \
    {                                                                                                   \
        const mpc_status s_ = (call);                                                                   \
        if (s_ != MPC_OK) { fail(text("lane %d: %s failed: %s", r, #call, mpc_last_error())); break; }   \
    }
        const int rows = me.end - me.begin;
        const size_t stripe_tiles = static_cast<size_t>(tiles_x) * rows;
        const int y0 = me.begin * 8, y1 = std::min(height, me.end * 8);
        const size_t band = row_bytes * (y1 - y0);                       // my rows of one frame
        if (hipSetDevice(me.device) != hipSuccess) { fail(text("lane %d: hipSetDevice failed", r)); return; }
        // The tile encodes follow each other on `stream`, on 7/8 of the CUs; the pulls, the interleave and the container job of a
        // step run on `side`, beside the next step's tile encode (as in the single-device frame pipeline).
        struct Workgroups {
            mpc_context* ctx;
            ~Workgroups() { (void)mpc_context_set_tile_encode_workgroups(ctx, 0); }
        } workgroups{me.ctx};
        {
            const int cus = mpc_context_max_waves(me.ctx) / 12;
            if (cus >= 16) (void)mpc_context_set_tile_encode_workgroups(me.ctx, cus - cus / 8);
        }
        // my rows of the frames of step s -> pinned memory (a few threads) -> the device, on the upload stream
        auto upload = [&](long s) -> bool {
            const int par = static_cast<int>(s & 1);
            const int first = static_cast<int>(s) * N, n_g = std::min(N, n_frames - first);
            uint8_t* pinned = me.h_rgb[par];
            std::vector<std::future<void>> parts;
            for (int f = 0; f < n_g; ++f) {
                const uint8_t* src = rgb_frames[first + f] + row_bytes * y0;
                uint8_t* dst = pinned + band * f;
                parts.push_back(std::async(std::launch::async, [=] { std::memcpy(dst, src, band); }));
            }
            for (auto& p : parts) p.get();
            for (int f = 0; f < n_g; ++f)
                if (hipMemcpyAsync(me.d_rgb[par] + frame_bytes * f + row_bytes * y0, pinned + band * f, band, hipMemcpyHostToDevice, me.up_stream) != hipSuccess)
                    return false;
            return hipEventRecord(me.uploaded[par], me.up_stream) == hipSuccess;
        };
        for (long s = 0; s <= steps + 1; ++s) {                          // two more turns drain the pipeline
            if (failed) break;
            if (s < steps) {
                const int par = static_cast<int>(s & 1);
                const int first = static_cast<int>(s) * N, n_g = std::min(N, n_frames - first);
                // the pinned buffer and d_rgb[par] were last used by step s - 2: its encode has been enqueued on my stream, and the
                // copy below is ordered behind it by the event wait; the pinned buffer is free once that step's upload has completed
                if (s >= 2) LANE_HIP(hipEventSynchronize(me.uploaded[par]));
                if (s >= 2) LANE_HIP(hipStreamWaitEvent(me.up_stream, me.encoded[par], 0));
                if (!upload(s)) { fail(text("lane %d: upload failed: %s", r, hipGetErrorString(hipGetLastError()))); break; }
                // the stripes of step s - 2 (same buffers) must have been pulled by their owners before this encode overwrites them
                bool ok = true;
                if (s >= 2)
                    for (int o = 0; o < N && ok; ++o) {                  // o == r: my own side stream has interleaved my stripe
                        ok = await(lanes[static_cast<size_t>(o)].pulled_step, s - 2);
                        if (ok && hipStreamWaitEvent(me.stream, lanes[static_cast<size_t>(o)].pulled[par], 0) != hipSuccess) ok = false;
                    }
                if (!ok) { if (!failed) fail(text("lane %d: waiting for the pulls of step %ld failed", r, s - 2)); break; }
                LANE_HIP(hipStreamWaitEvent(me.stream, me.uploaded[par], 0));
                LANE_MPC(mpc_encode_batch_device(me.ctx, me.d_rgb[par], n_g, frame_bytes, width, height, row_bytes, me.begin, me.end, quant,
                                                 me.d_counts[par], me.d_choices[par], nullptr, nullptr, 0, me.stream));
                LANE_HIP(hipEventRecord(me.encoded[par], me.stream));
                me.encoded_step.store(s, std::memory_order_release);
                // as the owner of frame r of this step: every lane's stripe of it -> whole-frame records -> container job
                if (r < n_g) {
                    const int slot = static_cast<int>(s % 3);
                    LANE_HIP(hipStreamWaitEvent(me.side, me.encoded[par], 0));
                    for (int q = 0; q < N && ok; ++q) {
                        Lane& src = lanes[static_cast<size_t>(q)];
                        const size_t src_tiles = static_cast<size_t>(tiles_x) * (src.end - src.begin);
                        const uint16_t* part_counts = src.d_counts[par] + 3 * src_tiles * r;
                        const mpc_basis_choice* part_choices = src.d_choices[par] + 3 * static_cast<size_t>(K) * src_tiles * r;
                        if (q != r) {
                            ok = await(src.encoded_step, s) && hipStreamWaitEvent(me.side, src.encoded[par], 0) == hipSuccess &&
                                 hipMemcpyPeerAsync(me.d_part_counts[static_cast<size_t>(q)], me.device, part_counts, src.device,
                                                    sizeof(uint16_t) * 3 * src_tiles, me.side) == hipSuccess &&
                                 hipMemcpyPeerAsync(me.d_part_choices[static_cast<size_t>(q)], me.device, part_choices, src.device,
                                                    sizeof(mpc_basis_choice) * 3 * K * src_tiles, me.side) == hipSuccess;
                            part_counts = me.d_part_counts[static_cast<size_t>(q)];
                            part_choices = me.d_part_choices[static_cast<size_t>(q)];
                        }
                        if (ok && mpc_interleave_stripe_device(me.ctx, part_counts, part_choices, width, height, src.begin, src.end,
                                                               me.d_frame_counts[slot], me.d_frame_choices[slot], me.side) != MPC_OK)
                            ok = false;
                    }
                    if (!ok) { if (!failed) fail(text("lane %d: pulling the stripes of step %ld failed: %s", r, s, mpc_last_error())); break; }
                    LANE_HIP(hipEventRecord(me.pulled[par], me.side));
                    me.pulled_step.store(s, std::memory_order_release);
                    LANE_MPC(mpc_container_job_begin(me.ctx, slot, me.d_frame_counts[slot], me.d_frame_choices[slot], width, height, quant, me.side));
                } else {
                    LANE_HIP(hipEventRecord(me.pulled[par], me.side));               // nothing to pull: the event is there for the waiters
                    me.pulled_step.store(s, std::memory_order_release);
                }
                (void)stripe_tiles;
            }
            if (s >= 1 && s - 1 < steps) {                               // step s - 1: tables on the host, phase 2 and the copy enqueued
                const int first = static_cast<int>(s - 1) * N, n_g = std::min(N, n_frames - first);
                if (r < n_g) LANE_MPC(mpc_container_job_tables(me.ctx, static_cast<int>((s - 1) % 3)));
            }
            if (s >= 2 && s - 2 < steps) {                               // step s - 2: its container
                const int first = static_cast<int>(s - 2) * N, n_g = std::min(N, n_frames - first);
                if (r < n_g) LANE_MPC(mpc_container_job_collect(me.ctx, static_cast<int>((s - 2) % 3), &bytes[first + r], &nbytes[first + r]));
            }
        }
#undef LANE_HIP
#undef LANE_MPC
        (void)hipStreamSynchronize(me.stream);
        (void)hipStreamSynchronize(me.side);
        (void)hipStreamSynchronize(me.up_stream);
    };

    auto guarded_lane = [&](int r) {                                   // nothing may leave a lane's thread but a flag
        try {
            run_lane(r);
        } catch (const std::exception& e) {
            if (lanes[static_cast<size_t>(r)].error.empty()) lanes[static_cast<size_t>(r)].error = text("lane %d: %s", r, e.what());
            failed = true;
        } catch (...) {
            failed = true;
        }
    };
    if (!failed) {
        std::vector<std::thread> threads;
        try {
            for (int r = 1; r < N; ++r) threads.emplace_back(guarded_lane, r);
        } catch (...) {
            failed = true;                                              // a lane without a thread: the others give up at their next wait
        }
        guarded_lane(0);
        for (auto& t : threads) t.join();
    }
    if (failed)                                                         // jobs that were begun and never collected: their slots back to idle
        for (int r = 0; r < N; ++r)
            for (int slot = 0; slot < 3; ++slot) (void)mpc_container_job_cancel(ctxs[r], slot);
    std::string why;
    for (auto& l : lanes) {
        if (why.empty() && !l.error.empty()) why = l.error;
        release(l);
    }
    if (failed) {
        for (int f = 0; f < n_frames; ++f) { mpc_free(bytes[f]); bytes[f] = nullptr; nbytes[f] = 0; }
        mpc_set_error_text(why.empty() ? "multi-device encode failed" : why.c_str());
        return MPC_ERR_HIP;
    }
    return MPC_OK;
}
