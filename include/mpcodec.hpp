// mpcodec.hpp -- header-only C++ mirror of the reference's CompressionLib interface over the C ABI of
// mpcodec.h, for C++ callers that want the reference's names, argument meaning and error behaviour:
//
//   img::image<T>, img::rgb                      ImageHelper/inc/image.h:14-135
//   matching::BasisChoice                        CompressionLib/inc/MatchingPursuit.h:13-17
//   compressed::CompressionContext, createCompressionContext, encodeImage, decodeImage, calculatePSNR
//                                                CompressionLib/inc/CompressedImage.h:22-76
//   matching::CalcMPDynamic                      CompressionLib/inc/MatchingPursuit.h:22
//   huffman::huffmanEncode/Decode, runLengthEncode/Decode
//                                                CompressionLib/inc/Huffman.h:12-19
//
// Differences forced by the device boundary, and only these:
//   * the dictionary lives in the context (host + HBM copy); the DynamicDictionaryFunction closures of the
//     reference's encodeImage/CalcMPDynamic are replaced by a reference to the context that owns it;
//   * errors are thrown the way the reference throws them: `throw new std::range_error(...)` (a POINTER),
//     so code written against the reference (`catch (std::range_error* e)`) keeps working.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "mpcodec.h"

namespace img {

typedef unsigned char uchar;

typedef struct {
    uchar r;
    uchar g;
    uchar b;
} rgb;

// row-major image with row pointers, like the reference's (image.h:91-135)
template <class T>
class image {
public:
    image(const size_t width, const size_t height, const bool init = true) : w(width), h(height) {
        data = new T[w * h];
        access = new T*[h];
        for (size_t i = 0; i < h; i++) access[i] = data + (i * w);
        if (init) std::memset(static_cast<void*>(data), 0, w * h * sizeof(T));
    }
    ~image() {
        delete[] data;
        delete[] access;
    }
    image(const image&) = delete;
    image& operator=(const image&) = delete;
    size_t width() const { return w; }
    size_t height() const { return h; }
    T* data;
    T** access;

private:
    size_t w, h;
};

}  // namespace img

#ifndef imRef
#define imRef(im, x, y) (im->access[y][x])
#endif

namespace matching {

typedef mpc_basis_choice BasisChoice;   // { unsigned short deltaId; unsigned short intCoeff; }

}  // namespace matching

namespace compressed {

namespace detail {
[[noreturn]] inline void raise() { throw new std::range_error(mpc_last_error()); }
inline void check(mpc_status st) {
    if (st != MPC_OK) raise();
}
}  // namespace detail

struct ChannelContext {
    std::vector<double> Quant;      // K quantisation steps (the reference's math::Vector Quant)
};

struct CompressionContext {
    size_t K{32};
    size_t BlockSize{8};
    ChannelContext Y;
    ChannelContext U;
    ChannelContext V;
    mpc_context* handle{nullptr};   // dictionary on the host and in HBM
    CompressionContext() = default;
    CompressionContext(const CompressionContext&) = delete;
    CompressionContext& operator=(const CompressionContext&) = delete;
    ~CompressionContext() { mpc_context_destroy(handle); }
};

// createCompressionContext(K, blockSize, bppAllocation): device < 0 gives a host-only context (tables, entropy
// stage, decode) -- encodeImage on it throws, there is no CPU fallback
inline std::unique_ptr<CompressionContext> createCompressionContext(size_t K, size_t blockSize, double bppAllocation,
                                                                    int device = 0) {
    std::unique_ptr<CompressionContext> context = std::make_unique<CompressionContext>();
    detail::check(mpc_context_create(static_cast<int>(K), static_cast<int>(blockSize), bppAllocation, device, &context->handle));
    context->K = K;
    context->BlockSize = blockSize;
    std::vector<double> q(3 * K);
    detail::check(mpc_context_get_quant(context->handle, q.data()));
    context->Y.Quant.assign(q.begin(), q.begin() + K);
    context->U.Quant.assign(q.begin() + K, q.begin() + 2 * K);
    context->V.Quant.assign(q.begin() + 2 * K, q.end());
    return context;
}

// encodeImage(imgIn, K, blockSize, quantY, quantU, quantV, dynamicY, dynamicU, dynamicV, outputByteSize):
// the quant tables are taken from the arguments (Compression.cpp:104-110 overwrites them for "max")
inline std::unique_ptr<uint8_t[]> encodeImage(const img::image<img::rgb>* imgIn, const CompressionContext& context,
                                              const double quantY[], const double quantU[], const double quantV[],
                                              size_t& outputByteSize) {
    const size_t K = context.K;
    std::vector<double> q(3 * K);
    std::memcpy(q.data(), quantY, K * sizeof(double));
    std::memcpy(q.data() + K, quantU, K * sizeof(double));
    std::memcpy(q.data() + 2 * K, quantV, K * sizeof(double));
    uint8_t* bytes = nullptr;
    size_t n = 0;
    detail::check(mpc_encode_image(context.handle, reinterpret_cast<const uint8_t*>(imgIn->data),
                                   static_cast<int>(imgIn->width()), static_cast<int>(imgIn->height()), q.data(), &bytes, &n));
    std::unique_ptr<uint8_t[]> out = std::make_unique<uint8_t[]>(n);
    std::memcpy(out.get(), bytes, n);
    mpc_free(bytes);
    outputByteSize = n;
    return out;
}

// encodeImage for a sequence of equally sized frames (Compression.cpp:117/166 calls encodeImage once per input file):
// the host entropy stage of frame n overlaps the transfers and the device encode of frame n+1 (mpc_encode_images)
inline std::vector<std::vector<uint8_t>> encodeImages(const std::vector<const img::image<img::rgb>*>& frames,
                                                      const CompressionContext& context, const double quantY[],
                                                      const double quantU[], const double quantV[]) {
    const size_t K = context.K;
    std::vector<double> q(3 * K);
    std::memcpy(q.data(), quantY, K * sizeof(double));
    std::memcpy(q.data() + K, quantU, K * sizeof(double));
    std::memcpy(q.data() + 2 * K, quantV, K * sizeof(double));
    std::vector<std::vector<uint8_t>> out;
    if (frames.empty()) return out;
    std::vector<const uint8_t*> ptrs;
    for (const img::image<img::rgb>* f : frames) ptrs.push_back(reinterpret_cast<const uint8_t*>(f->data));
    std::vector<uint8_t*> bytes(frames.size(), nullptr);
    std::vector<size_t> sizes(frames.size(), 0);
    detail::check(mpc_encode_images(context.handle, ptrs.data(), static_cast<int>(frames.size()), static_cast<int>(frames[0]->width()),
                                    static_cast<int>(frames[0]->height()), q.data(), bytes.data(), sizes.data()));
    for (size_t i = 0; i < frames.size(); ++i) {
        out.emplace_back(bytes[i], bytes[i] + sizes[i]);
        mpc_free(bytes[i]);
    }
    return out;
}

// The reference builds a context from the stream header on every call (CompressedImage.cpp:783-790).  Here any
// device context of block size 8 decodes any stream (K and the tables come from the stream); without an explicit
// one a process-wide context on device 0 is created on first use.  Reconstruction runs on the device only.
inline std::unique_ptr<img::image<img::rgb>> decodeImage(const uint8_t bytes[], size_t byteSize,
                                                         const CompressionContext* context = nullptr) {
    const mpc_context* handle = context ? context->handle : nullptr;
    if (!handle) {
        static const std::unique_ptr<CompressionContext> decoder = createCompressionContext(32, 8, 3.5, 0);
        handle = decoder->handle;
    }
    uint8_t* rgb = nullptr;
    int w = 0, h = 0;
    detail::check(mpc_decode_image(handle, bytes, byteSize, &rgb, &w, &h));
    std::unique_ptr<img::image<img::rgb>> imgOut =
        std::make_unique<img::image<img::rgb>>(static_cast<size_t>(w), static_cast<size_t>(h), false);
    std::memcpy(static_cast<void*>(imgOut->data), rgb, static_cast<size_t>(w) * h * 3);
    mpc_free(rgb);
    return imgOut;
}

inline double calculatePSNR(const img::image<img::rgb>* original, const img::image<img::rgb>* decoded) {
    return mpc_psnr(reinterpret_cast<const uint8_t*>(original->data), reinterpret_cast<const uint8_t*>(decoded->data),
                    static_cast<int>(original->width()), static_cast<int>(original->height()));
}

// The "-s" mode of Compression.cpp:200-302 as an object: one generator for the whole file list, addImage per
// file, report() = the text the reference writes to its output file.
class PatchStatistics {
public:
    PatchStatistics(CompressionContext& context, unsigned seed) { detail::check(mpc_patch_stats_create(context.handle, seed, &handle_)); }
    ~PatchStatistics() { mpc_patch_stats_destroy(handle_); }
    PatchStatistics(const PatchStatistics&) = delete;
    PatchStatistics& operator=(const PatchStatistics&) = delete;
    void addImage(const img::image<img::rgb>* image, int patchesPerImage) {
        detail::check(mpc_patch_stats_add_image(handle_, reinterpret_cast<const uint8_t*>(image->data),
                                                static_cast<int>(image->width()), static_cast<int>(image->height()), patchesPerImage));
    }
    std::string report() const {
        char* text = nullptr;
        size_t n = 0;
        detail::check(mpc_patch_stats_report(handle_, &text, &n));
        std::string out(text, n);
        mpc_free(text);
        return out;
    }

private:
    mpc_patch_stats* handle_ = nullptr;
};

}  // namespace compressed

namespace matching {

// CalcMPDynamic(K, quantization, results, input, dynamicDictionary) -> count; `channel` selects whose detail
// dictionary the context applies (0 Y, 1 U, 2 V), as the reference's closure did
inline int CalcMPDynamic(const compressed::CompressionContext& context, int channel, const double quantization[],
                         std::vector<BasisChoice>& results, const double input[]) {
    if (results.size() < context.K) results.resize(context.K);
    int count = 0;
    compressed::detail::check(mpc_calc_mp(context.handle, channel, quantization, input, results.data(), &count));
    return count;
}

}  // namespace matching

namespace huffman {

inline std::vector<uint16_t> runLengthEncode(const std::vector<uint16_t>& data) {
    uint16_t* out = nullptr;
    size_t n = 0;
    compressed::detail::check(mpc_rle_encode(data.data(), data.size(), &out, &n));
    std::vector<uint16_t> v(out, out + n);
    mpc_free(out);
    return v;
}

inline std::vector<uint16_t> runLengthDecode(const std::vector<uint16_t>& data) {
    uint16_t* out = nullptr;
    size_t n = 0;
    compressed::detail::check(mpc_rle_decode(data.data(), data.size(), &out, &n));
    std::vector<uint16_t> v(out, out + n);
    mpc_free(out);
    return v;
}

inline std::unique_ptr<uint8_t[]> huffmanEncode(const std::vector<uint16_t>& data, size_t& outputLength) {
    uint8_t* bytes = nullptr;
    size_t n = 0;
    compressed::detail::check(mpc_huffman_encode(data.data(), data.size(), &bytes, &n));
    std::unique_ptr<uint8_t[]> out = std::make_unique<uint8_t[]>(n ? n : 1);
    std::memcpy(out.get(), bytes, n);
    mpc_free(bytes);
    outputLength = n;
    return out;
}

inline std::vector<uint16_t> huffmanDecode(const uint8_t encoded[], size_t compressedLength) {
    uint16_t* out = nullptr;
    size_t n = 0;
    compressed::detail::check(mpc_huffman_decode(encoded, compressedLength, &out, &n));   // throws std::range_error*
    std::vector<uint16_t> v(out, out + n);
    mpc_free(out);
    return v;
}

}  // namespace huffman
