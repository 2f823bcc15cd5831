// compressionlib_dropin.cpp -- product: CompressionLib's encode / decode entry points, with the reference's exact signatures,
// over libmpcodec.so (the MI355X tile encoder).  A maintainer of mnesbit/ImageExperiments compiles THIS file in the
// CompressionLib project in place of src/CompressedImage.cpp and src/MatchingPursuit.cpp (it includes the reference's own
// headers, so every type and signature is the reference's by construction), links -lmpcodec, and Compression.cpp -- and any
// other caller -- builds and links unchanged.
//
//   compressed::createQuantizationTables        CompressionLib/inc/CompressedImage.h:15-20
//   compressed::createCompressionContext[Fast]  :54-55     dictionary built by libmpcodec (bit-identical to the reference's
//                                                          double path), copied into the reference's context struct
//   compressed::encodeImage[Fast]               :59-73     tile encode on the GPU, entropy stage on the host, same bytes
//   compressed::decodeImage[Fast]               :75-76     container parsed on the host, tiles reconstructed on the GPU
//   compressed::calculatePSNR                   :57
//   matching::CalcMPDynamic[Fast]               CompressionLib/inc/MatchingPursuit.h:22-23   (Compression.cpp -s mode)
//   matching::FromCoeffsDynamic[Fast]           :25-26
//
// The dictionary crosses the reference's interface only as an opaque std::function ("dynamic dictionary").  Contexts made by
// the factory here carry closures of a type this file recognises (std::function::target), which lead straight to the
// mpc_context that holds the dictionary in HBM.  A foreign closure -- e.g. one made by the reference's own factory -- is
// probed (dyn(0, {}) is the base dictionary, dyn(1, {i}) appends DetailBasis[i]; in full the first time a
// std::function object is seen, a spot check on later calls: `identify`): if it is the standard dictionary of that channel the
// same device path is used, otherwise the call fails like the reference fails, with a thrown std::range_error*.
//
// The Fast names (Eigen, float) run the float flavour of the device path (mpc_context_set_fast): the reference's Fast
// statements in float on the dictionary rounded to float -- equivalent to the double path in PSNR and size, identical to
// oracle/mpo_fast.c, NOT pinned to the reference's own float results (those depend on Eigen's summation order and on its float
// eigensolver; see include/mpcodec.h).  MPC_FAST_EXACT=1 serves them from the double path instead.  Build with
// -DMPC_DROPIN_NO_EIGEN in a tree without Eigen to leave them out.
#include "CompressedImage.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

#include <mpcodec.h>

namespace {

[[noreturn]] void raise() { throw new std::range_error(mpc_last_error()); }      // the reference throws pointers
void check(mpc_status st) {
    if (st != MPC_OK) raise();
}

int device_ordinal() {
    const char* v = std::getenv("MPC_DEVICE");
    return v && *v ? std::atoi(v) : 0;
}

// MPC_DEVICES=0,1,2,...: encodeImage stripes every frame's tile rows over these devices (mpc_encode_images_multi; a device may be
// named more than once: one lane each).  Unset or a single entry: one device, MPC_DEVICE (default 0).
std::vector<int> device_list() {
    std::vector<int> out;
    if (const char* v = std::getenv("MPC_DEVICES")) {
        const char* p = v;
        while (*p) {
            char* end = nullptr;
            const long d = std::strtol(p, &end, 10);
            if (end == p) break;
            out.push_back(static_cast<int>(d));
            p = *end == ',' ? end + 1 : end;
        }
    }
    if (out.empty()) out.push_back(device_ordinal());
    return out;
}

// one mpc_context per (K, block size): the dictionary does not depend on the bit allocation, the quantisers travel as
// arguments of every call (Compression.cpp:104-110 overwrites them in place for "max")
struct Handles {
    std::mutex lock;
    std::map<std::pair<size_t, size_t>, mpc_context*> by_shape;
    // fast: the float flavour (its own context: the flag is per context; the device dictionary is shared anyway)
    std::map<std::pair<size_t, size_t>, std::vector<mpc_context*>> lanes_by_shape;    // MPC_DEVICES: one context per lane
    mpc_context* get(size_t K, size_t blockSize, bool fast = false) {
        std::lock_guard<std::mutex> hold(lock);
        mpc_context*& h = by_shape[{2 * K + (fast ? 1 : 0), blockSize}];
        if (!h) {
            check(mpc_context_create(static_cast<int>(K), static_cast<int>(blockSize), 3.5, device_list()[0], &h));
            if (fast) check(mpc_context_set_fast(h, 1));
        }
        return h;
    }
    const std::vector<mpc_context*>& lanes(size_t K, size_t blockSize, bool fast = false) {
        std::lock_guard<std::mutex> hold(lock);
        std::vector<mpc_context*>& v = lanes_by_shape[{2 * K + (fast ? 1 : 0), blockSize}];
        if (v.empty())
            for (int device : device_list()) {
                mpc_context* h = nullptr;
                check(mpc_context_create(static_cast<int>(K), static_cast<int>(blockSize), 3.5, device, &h));
                v.push_back(h);
                if (fast) check(mpc_context_set_fast(h, 1));
            }
        return v;
    }
    ~Handles() {
        for (auto& kv : by_shape) mpc_context_destroy(kv.second);
        for (auto& kv : lanes_by_shape)
            for (mpc_context* h : kv.second) mpc_context_destroy(h);
    }
};
Handles& handles() {
    static Handles h;
    return h;
}

// host copy of the standard dictionary (base rows, rows per block, detail rows per channel)
struct HostDictionary {
    int num_base = 0, detail_rows = 0, n = 0;
    std::vector<double> base;
    std::vector<int32_t> rows, offset;
    std::vector<double> detail[3];
};
const HostDictionary& host_dictionary(size_t blockSize) {
    static std::mutex lock;
    static std::map<size_t, HostDictionary> cache;
    std::lock_guard<std::mutex> hold(lock);
    HostDictionary& d = cache[blockSize];
    if (d.num_base == 0) {
        mpc_context* h = nullptr;
        check(mpc_context_create(1, static_cast<int>(blockSize), 0.0, -1, &h));           // host-only: tables and dictionary
        d.num_base = mpc_context_num_base(h);
        d.detail_rows = mpc_context_detail_rows(h);
        d.n = static_cast<int>(blockSize * blockSize);
        d.base.resize(static_cast<size_t>(d.num_base) * d.n);
        d.rows.resize(d.num_base);
        for (auto& ch : d.detail) ch.resize(static_cast<size_t>(d.detail_rows) * d.n);
        check(mpc_context_get_dictionary(h, d.base.data(), d.rows.data(), d.detail[0].data(), d.detail[1].data(), d.detail[2].data()));
        mpc_context_destroy(h);
        d.offset.assign(d.num_base + 1, 0);
        for (int i = 0; i < d.num_base; ++i) d.offset[i + 1] = d.offset[i] + d.rows[i];
    }
    return d;
}

// compressed::dynamicBasis (CompressedImage.cpp:212-250): base rows, then DetailBasis[choice] of every earlier choice below
// the base count, in order, repeats included.  Row-major doubles.
std::vector<double> dynamic_rows(const HostDictionary& d, int channel, int prevCoeffs, const std::vector<matching::BasisChoice>& choices,
                                 size_t* rows_out) {
    std::vector<double> m(d.base);
    int id = 0;
    for (int i = 0; i < prevCoeffs && i < static_cast<int>(choices.size()); ++i) {
        const unsigned z = choices[i].deltaId;
        id = i == 0 ? static_cast<int>(z) : id + static_cast<int>((z >> 1) ^ (0u - (z & 1u)));
        if (id >= 0 && id < d.num_base) {
            const double* first = d.detail[channel].data() + static_cast<size_t>(d.offset[id]) * d.n;
            m.insert(m.end(), first, first + static_cast<size_t>(d.rows[id]) * d.n);
        }
    }
    *rows_out = m.size() / d.n;
    return m;
}

// The closure type of contexts made here: callable like the reference's, and recognisable.
struct StandardDynamic {
    size_t K, blockSize;
    int channel;
    math::Matrix operator()(int prevCoeffs, const std::vector<matching::BasisChoice>& choices) const {
        const HostDictionary& d = host_dictionary(blockSize);
        size_t rows = 0;
        const std::vector<double> m = dynamic_rows(d, channel, prevCoeffs, choices, &rows);
        return math::Matrix(rows, static_cast<size_t>(d.n), m.data());
    }
};

// Does `dyn` return the standard dictionary of channel `ch` for these base atoms?  dyn(0, {}) is the base dictionary,
// dyn(1, {i}) appends DetailBasis[i].
bool probe(const matching::DynamicDictionaryFunction& dyn, const HostDictionary& d, int ch, const int* atoms, int n_atoms, bool base_too) {
    if (base_too) {
        const std::vector<matching::BasisChoice> none;
        const math::Matrix base = dyn(0, none);
        if (base.Rows() != static_cast<size_t>(d.num_base) || base.Columns() != static_cast<size_t>(d.n)) return false;
        if (std::memcmp(base.Data(), d.base.data(), d.base.size() * sizeof(double)) != 0) return false;
    }
    for (int k = 0; k < n_atoms; ++k) {
        const int i = atoms[k];
        std::vector<matching::BasisChoice> one(1);
        one[0].deltaId = static_cast<unsigned short>(i);
        one[0].intCoeff = 0;
        const math::Matrix m = dyn(1, one);
        if (m.Rows() != static_cast<size_t>(d.num_base + d.rows[i]) || m.Columns() != static_cast<size_t>(d.n) ||
            std::memcmp(m.Data() + d.base.size(), d.detail[ch].data() + static_cast<size_t>(d.offset[i]) * d.n,
                        static_cast<size_t>(d.rows[i]) * d.n * sizeof(double)) != 0)
            return false;
    }
    return true;
}

// Which channel of the standard dictionary does a closure stand for?  -1 = none.  A closure of the factory here is recognised
// by its type.  A foreign one is probed in full the first time it is seen (base + all 510 detail blocks against each channel);
// the verdict is remembered for that std::function object (its address and target type), and a later call with the same object
// re-checks only the base dictionary and three detail blocks -- an address can be reused by another closure, a whole different
// dictionary behind the same object and type cannot hide from that.
int identify(const matching::DynamicDictionaryFunction& dyn, size_t blockSize) {
    if (const StandardDynamic* mine = dyn.target<StandardDynamic>()) return mine->blockSize == blockSize ? mine->channel : -1;
    if (!dyn) return -1;
    const HostDictionary& d = host_dictionary(blockSize);
    struct Seen { const void* object; size_t type; size_t blockSize; int channel; };
    static std::mutex lock;
    static std::vector<Seen> seen;
    const size_t type = dyn.target_type().hash_code();
    {
        std::lock_guard<std::mutex> hold(lock);
        for (const Seen& s : seen)
            if (s.object == &dyn && s.type == type && s.blockSize == blockSize) {
                const int spot[3] = {0, d.num_base / 2, d.num_base - 1};
                if (s.channel >= 0 && probe(dyn, d, s.channel, spot, 3, true)) return s.channel;
                break;                                    // not what it was: probe in full again
            }
    }
    std::vector<int> all(static_cast<size_t>(d.num_base));
    for (int i = 0; i < d.num_base; ++i) all[static_cast<size_t>(i)] = i;
    int channel = -1;
    if (probe(dyn, d, 0, nullptr, 0, true))
        for (int ch = 0; ch < 3 && channel < 0; ++ch)
            if (probe(dyn, d, ch, all.data(), d.num_base, false)) channel = ch;
    std::lock_guard<std::mutex> hold(lock);
    bool updated = false;
    for (Seen& s : seen)
        if (s.object == &dyn && s.blockSize == blockSize) { s.type = type; s.channel = channel; updated = true; }
    if (!updated) {
        if (seen.size() >= 64) seen.erase(seen.begin());
        seen.push_back(Seen{&dyn, type, blockSize, channel});
    }
    return channel;
}

void require_standard(const matching::DynamicDictionaryFunction& y, const matching::DynamicDictionaryFunction& u,
                      const matching::DynamicDictionaryFunction& v, size_t blockSize) {
    if (identify(y, blockSize) != 0 || identify(u, blockSize) != 1 || identify(v, blockSize) != 2)
        throw new std::range_error("encodeImage: the dynamic dictionaries are not createCompressionContext's (only the standard "
                                   "segment + KLT dictionary is resident on the device)");
}

std::unique_ptr<uint8_t[]> encode(const img::image<img::rgb>* imgIn, size_t K, size_t blockSize, const double* qY, const double* qU,
                                  const double* qV, size_t& outputByteSize, bool fast = false) {
    std::vector<double> q(3 * K);
    std::memcpy(q.data(), qY, K * sizeof(double));
    std::memcpy(q.data() + K, qU, K * sizeof(double));
    std::memcpy(q.data() + 2 * K, qV, K * sizeof(double));
    uint8_t* bytes = nullptr;
    size_t n = 0;
    if (device_list().size() > 1 && (imgIn->height() + 7) / 8 >= device_list().size()) {          // the frame's tile rows over several GPUs
        const std::vector<mpc_context*>& lanes = handles().lanes(K, blockSize, fast);
        const uint8_t* frame = reinterpret_cast<const uint8_t*>(imgIn->data);
        check(mpc_encode_images_multi(lanes.data(), static_cast<int>(lanes.size()), &frame, 1, static_cast<int>(imgIn->width()),
                                      static_cast<int>(imgIn->height()), q.data(), &bytes, &n));
    } else {
        check(mpc_encode_image(handles().get(K, blockSize, fast), reinterpret_cast<const uint8_t*>(imgIn->data), static_cast<int>(imgIn->width()),
                               static_cast<int>(imgIn->height()), q.data(), &bytes, &n));
    }
    std::unique_ptr<uint8_t[]> out = std::make_unique<uint8_t[]>(n ? n : 1);
    std::memcpy(out.get(), bytes, n);
    mpc_free(bytes);
    outputByteSize = n;
    return out;
}

std::unique_ptr<img::image<img::rgb>> decode(const uint8_t bytes[], size_t byteSize, bool fast = false) {
    if (byteSize < 14) throw new std::range_error("Invalid input data");
    const size_t blockSize = bytes[13];                   // header: magic, width, height (3 x u32), K (u8), block size (u8)
    uint8_t* rgb = nullptr;
    int w = 0, h = 0;
    check(mpc_decode_image(handles().get(32, blockSize ? blockSize : 8, fast), bytes, byteSize, &rgb, &w, &h));
    std::unique_ptr<img::image<img::rgb>> out = std::make_unique<img::image<img::rgb>>(static_cast<size_t>(w), static_cast<size_t>(h), false);
    std::memcpy(static_cast<void*>(out->data), rgb, static_cast<size_t>(w) * h * 3);
    mpc_free(rgb);
    return out;
}

}  // namespace

namespace compressed {

void createQuantizationTables(const size_t K, const size_t blockSize, const double bppAllocation, math::Vector& quantY,
                              math::Vector& quantU, math::Vector& quantV) {
    mpc_context* h = nullptr;
    check(mpc_context_create(static_cast<int>(K), static_cast<int>(blockSize), bppAllocation, -1, &h));
    std::vector<double> q(3 * K);
    const mpc_status st = mpc_context_get_quant(h, q.data());
    mpc_context_destroy(h);
    check(st);
    quantY = math::Vector(K, q.data());
    quantU = math::Vector(K, q.data() + K);
    quantV = math::Vector(K, q.data() + 2 * K);
}

std::unique_ptr<CompressionContext> createCompressionContext(size_t K, size_t blockSize, double bppAllocation) {
    std::unique_ptr<CompressionContext> context = std::make_unique<CompressionContext>();
    context->K = K;
    context->BlockSize = blockSize;
    createQuantizationTables(K, blockSize, bppAllocation, context->Y.Quant, context->U.Quant, context->V.Quant);
    const HostDictionary& d = host_dictionary(blockSize);
    context->BaseDict = math::Matrix(static_cast<size_t>(d.num_base), static_cast<size_t>(d.n), d.base.data());
    ChannelContext* channels[3] = {&context->Y, &context->U, &context->V};
    for (int ch = 0; ch < 3; ++ch) {
        for (int i = 0; i < d.num_base; ++i)
            channels[ch]->DetailBasis.emplace_back(static_cast<size_t>(d.rows[i]), static_cast<size_t>(d.n),
                                                   d.detail[ch].data() + static_cast<size_t>(d.offset[i]) * d.n);
        channels[ch]->Dynamic = StandardDynamic{K, blockSize, ch};
    }
    return context;
}

double calculatePSNR(const img::image<img::rgb>* original, const img::image<img::rgb>* decoded) {
    return mpc_psnr(reinterpret_cast<const uint8_t*>(original->data), reinterpret_cast<const uint8_t*>(decoded->data),
                    static_cast<int>(original->width()), static_cast<int>(original->height()));
}

std::unique_ptr<uint8_t[]> encodeImage(const img::image<img::rgb>* imgIn, const size_t K, const size_t blockSize, const double quantY[],
                                       const double quantU[], const double quantV[], const matching::DynamicDictionaryFunction& dynamicY,
                                       const matching::DynamicDictionaryFunction& dynamicU, const matching::DynamicDictionaryFunction& dynamicV,
                                       size_t& outputByteSize) {
    require_standard(dynamicY, dynamicU, dynamicV, blockSize);
    return encode(imgIn, K, blockSize, quantY, quantU, quantV, outputByteSize);
}

std::unique_ptr<img::image<img::rgb>> decodeImage(const uint8_t bytes[], size_t byteSize) { return decode(bytes, byteSize); }

}  // namespace compressed

namespace matching {

int CalcMPDynamic(int K, const double quantization[], std::vector<BasisChoice>& results, const math::Vector& input,
                  const DynamicDictionaryFunction& dynamicDictionary) {
    const size_t n = input.Length();
    size_t blockSize = 1;
    while (blockSize * blockSize < n) ++blockSize;
    const int channel = identify(dynamicDictionary, blockSize);
    if (channel < 0) throw new std::range_error("CalcMPDynamic: not a dynamic dictionary of createCompressionContext");
    if (results.size() < static_cast<size_t>(K)) results.resize(static_cast<size_t>(K));
    static_assert(sizeof(BasisChoice) == sizeof(mpc_basis_choice), "BasisChoice is two unsigned shorts");
    int count = 0;
    check(mpc_calc_mp(handles().get(static_cast<size_t>(K), blockSize), channel, quantization, input.Data(),
                      reinterpret_cast<mpc_basis_choice*>(results.data()), &count));
    return count;
}

// MatchingPursuit.cpp:109-128: sum of coefficient * row over the recorded steps, rows resolved in the dictionary built from
// ALL of them (one tile: host arithmetic; whole frames go through decodeImage on the device)
math::Vector FromCoeffsDynamic(int K, const double quantization[], const std::vector<BasisChoice>& coeffs,
                               const DynamicDictionaryFunction& dynamicDictionary) {
    const math::Matrix dictionary = dynamicDictionary(static_cast<int>(coeffs.size()), coeffs);
    math::Vector results(dictionary.Columns());
    int choice = 0;
    for (size_t i = 0; i < coeffs.size() && i < static_cast<size_t>(K); ++i) {
        const unsigned z = coeffs[i].deltaId, c = coeffs[i].intCoeff;
        choice = i == 0 ? static_cast<int>(z) : choice + static_cast<int>((z >> 1) ^ (0u - (z & 1u)));
        const double coeff = quantization[i] * static_cast<double>(static_cast<int>((c >> 1) ^ (0u - (c & 1u))));
        if (choice < 0 || static_cast<size_t>(choice) >= dictionary.Rows()) throw new std::range_error("Invalid input data");
        const double* row = dictionary.Data() + static_cast<size_t>(choice) * dictionary.Columns();
        for (size_t j = 0; j < dictionary.Columns(); ++j) {
            const double term = row[j] * coeff;
            results[j] = results[j] + term;
        }
    }
    return results;
}

}  // namespace matching

#ifndef MPC_DROPIN_NO_EIGEN
// ---- the names Compression.cpp actually calls (:98-123, :147-170): Eigen types at the interface, the same device path ----
namespace {
struct StandardDynamicFast {
    size_t K, blockSize;
    int channel;
    Eigen::MatrixXf operator()(int prevCoeffs, const std::vector<matching::BasisChoice>& choices) const {
        const HostDictionary& d = host_dictionary(blockSize);
        size_t rows = 0;
        const std::vector<double> m = dynamic_rows(d, channel, prevCoeffs, choices, &rows);
        Eigen::MatrixXf out(static_cast<Eigen::Index>(rows), static_cast<Eigen::Index>(d.n));
        for (size_t r = 0; r < rows; ++r)
            for (int c = 0; c < d.n; ++c) out(static_cast<Eigen::Index>(r), c) = static_cast<float>(m[r * d.n + c]);
        return out;
    }
};
int identify_fast(const matching::DynamicDictionaryFunctionFast& dyn, size_t blockSize) {
    const StandardDynamicFast* mine = dyn.target<StandardDynamicFast>();
    return mine && mine->blockSize == blockSize ? mine->channel : -1;
}
bool fast_flavour() {
    const char* v = std::getenv("MPC_FAST_EXACT");
    return !(v && *v && std::atoi(v) != 0);
}
std::vector<double> widen(const Eigen::VectorXf& v) {
    std::vector<double> out(static_cast<size_t>(v.size()));
    for (Eigen::Index i = 0; i < v.size(); ++i) out[static_cast<size_t>(i)] = static_cast<double>(v[i]);
    return out;
}
}  // namespace

namespace compressed {

std::unique_ptr<CompressionContextFast> createCompressionContextFast(size_t K, size_t blockSize, double bppAllocation) {
    std::unique_ptr<CompressionContextFast> context = std::make_unique<CompressionContextFast>();
    context->K = K;
    context->BlockSize = blockSize;
    math::Vector q[3];
    createQuantizationTables(K, blockSize, bppAllocation, q[0], q[1], q[2]);
    const HostDictionary& d = host_dictionary(blockSize);
    context->BaseDict = Eigen::MatrixXf(static_cast<Eigen::Index>(d.num_base), static_cast<Eigen::Index>(d.n));
    for (int r = 0; r < d.num_base; ++r)
        for (int c = 0; c < d.n; ++c) context->BaseDict(r, c) = static_cast<float>(d.base[static_cast<size_t>(r) * d.n + c]);
    ChannelContextFast* channels[3] = {&context->Y, &context->U, &context->V};
    for (int ch = 0; ch < 3; ++ch) {
        channels[ch]->Quant = Eigen::VectorXf(static_cast<Eigen::Index>(K));
        for (size_t i = 0; i < K; ++i) channels[ch]->Quant[static_cast<Eigen::Index>(i)] = static_cast<float>(q[ch][i]);
        for (int i = 0; i < d.num_base; ++i) {
            Eigen::MatrixXf block(static_cast<Eigen::Index>(d.rows[i]), static_cast<Eigen::Index>(d.n));
            for (int r = 0; r < d.rows[i]; ++r)
                for (int c = 0; c < d.n; ++c)
                    block(r, c) = static_cast<float>(d.detail[ch][(static_cast<size_t>(d.offset[i]) + r) * d.n + c]);
            channels[ch]->DetailBasis.push_back(block);
        }
        channels[ch]->Dynamic = StandardDynamicFast{K, blockSize, ch};
    }
    return context;
}

std::unique_ptr<uint8_t[]> encodeImageFast(const img::image<img::rgb>* imgIn, const size_t K, const size_t blockSize,
                                           const Eigen::VectorXf& quantY, const Eigen::VectorXf& quantU, const Eigen::VectorXf& quantV,
                                           const matching::DynamicDictionaryFunctionFast& dynamicY,
                                           const matching::DynamicDictionaryFunctionFast& dynamicU,
                                           const matching::DynamicDictionaryFunctionFast& dynamicV, size_t& outputByteSize) {
    if (identify_fast(dynamicY, blockSize) != 0 || identify_fast(dynamicU, blockSize) != 1 || identify_fast(dynamicV, blockSize) != 2)
        throw new std::range_error("encodeImageFast: the dynamic dictionaries are not createCompressionContextFast's");
    const std::vector<double> qY = widen(quantY), qU = widen(quantU), qV = widen(quantV);
    return encode(imgIn, K, blockSize, qY.data(), qU.data(), qV.data(), outputByteSize, fast_flavour());
}

std::unique_ptr<img::image<img::rgb>> decodeImageFast(const uint8_t bytes[], size_t byteSize) { return decode(bytes, byteSize, fast_flavour()); }

}  // namespace compressed

namespace matching {

int CalcMPDynamicFast(int K, const Eigen::VectorXf& quantization, std::vector<BasisChoice>& results, const Eigen::VectorXf& input,
                      const DynamicDictionaryFunctionFast& dynamicDictionary) {
    size_t blockSize = 1;
    while (blockSize * blockSize < static_cast<size_t>(input.size())) ++blockSize;
    const int channel = identify_fast(dynamicDictionary, blockSize);
    if (channel < 0) throw new std::range_error("CalcMPDynamicFast: not a dynamic dictionary of createCompressionContextFast");
    if (results.size() < static_cast<size_t>(K)) results.resize(static_cast<size_t>(K));
    const std::vector<double> q = widen(quantization), in = widen(input);
    int count = 0;
    check(mpc_calc_mp(handles().get(static_cast<size_t>(K), blockSize, fast_flavour()), channel, q.data(), in.data(),
                      reinterpret_cast<mpc_basis_choice*>(results.data()), &count));
    return count;
}

Eigen::VectorXf FromCoeffsDynamicFast(int K, const Eigen::VectorXf& quantization, const std::vector<BasisChoice>& coeffs,
                                      const DynamicDictionaryFunctionFast& dynamicDictionary) {
    const Eigen::MatrixXf dictionary = dynamicDictionary(static_cast<int>(coeffs.size()), coeffs);
    Eigen::VectorXf results(dictionary.cols());
    for (Eigen::Index j = 0; j < dictionary.cols(); ++j) results[j] = 0.0f;
    int choice = 0;
    for (size_t i = 0; i < coeffs.size() && i < static_cast<size_t>(K); ++i) {
        const unsigned z = coeffs[i].deltaId, c = coeffs[i].intCoeff;
        choice = i == 0 ? static_cast<int>(z) : choice + static_cast<int>((z >> 1) ^ (0u - (z & 1u)));
        const float coeff = quantization[static_cast<Eigen::Index>(i)] * static_cast<float>(static_cast<int>((c >> 1) ^ (0u - (c & 1u))));
        if (choice < 0 || choice >= dictionary.rows()) throw new std::range_error("Invalid input data");
        for (Eigen::Index j = 0; j < dictionary.cols(); ++j) results[j] += dictionary(choice, j) * coeff;
    }
    return results;
}

}  // namespace matching
#endif  // MPC_DROPIN_NO_EIGEN
