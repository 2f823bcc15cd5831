// reference_interface.h -- TEST SCAFFOLDING, not product code and not a build of the reference.
// The GPU box has no /root/reference, and CompressionLib's own headers cannot be compiled here anyway (they include Eigen and
// <format>, neither present).  To compile dropin/compressionlib_dropin.cpp and a caller written like Compression.cpp in the
// tests, this file DECLARES the part of CompressionLib's public interface that those two touch -- names, signatures and member
// names as in CompressionLib/inc/{CompressedImage,MatchingPursuit}.h -- over the value types math::Vector / math::Matrix /
// img::image, which come from one of two places:
//   * default: restated below with the smallest bodies that make them usable (row-major double storage), as in
//     SimpleMatrix/inc/{mathvector,mathmatrix}.h and ImageHelper/inc/image.h;
//   * -DMPC_TEST_REAL_REFERENCE_HEADERS -I <reference root>: the reference's REAL headers of those three, which compile here
//     unmodified (tests/test_dropin.py: test_dropin_builds_against_the_reference_s_own_value_types; their member functions are
//     then linked from SimpleMatrix/src/{mathmatrix,mathvector}.cpp compiled in place).  Only CompressedImage.h /
//     MatchingPursuit.h stay declarations (Eigen).
// The one-line headers under tests/cpp/refstub/<project>/inc/ forward here, so the drop-in's `#include "CompressedImage.h"` resolves.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#ifdef MPC_TEST_REAL_REFERENCE_HEADERS
#include "SimpleMatrix/inc/mathmatrix.h"
#include "SimpleMatrix/inc/mathvector.h"
#include "ImageHelper/inc/image.h"
#else
namespace math {
class Vector {
public:
    Vector() = default;
    Vector(size_t length, const double* vectordata = 0) : v_(length, 0.0) {
        if (vectordata) std::memcpy(v_.data(), vectordata, length * sizeof(double));
    }
    const double& operator[](size_t i) const { return v_[i]; }
    double& operator[](size_t i) { return v_[i]; }
    size_t Length() const { return v_.size(); }
    double* Data() { return v_.data(); }
    const double* Data() const { return v_.data(); }
private:
    std::vector<double> v_;
};
class Matrix {
public:
    Matrix() = default;
    Matrix(size_t rows, size_t columns, const double* arraydata = 0) : m_(rows), n_(columns), v_(rows * columns, 0.0) {
        if (arraydata) std::memcpy(v_.data(), arraydata, rows * columns * sizeof(double));
    }
    size_t Rows() const { return m_; }
    size_t Columns() const { return n_; }
    double* Data() { return v_.data(); }
    const double* Data() const { return v_.data(); }
private:
    size_t m_ = 0, n_ = 0;
    std::vector<double> v_;
};
}  // namespace math

namespace img {
typedef unsigned char uchar;
typedef struct { uchar r; uchar g; uchar b; } rgb;
template <class T>
class image {
public:
    image(const size_t width, const size_t height, const bool init = true) : w(width), h(height) {
        data = new T[w * h];
        access = new T*[h];
        for (size_t i = 0; i < h; i++) access[i] = data + (i * w);
        if (init) std::memset(static_cast<void*>(data), 0, w * h * sizeof(T));
    }
    ~image() { delete[] data; delete[] access; }
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
#define imRef(im, x, y) (im->access[y][x])
#endif  // MPC_TEST_REAL_REFERENCE_HEADERS

namespace Eigen {                                   // just enough of VectorXf / MatrixXf for the Fast signatures to exist
typedef std::ptrdiff_t Index;
class VectorXf {
public:
    VectorXf() = default;
    explicit VectorXf(Index n) : v_(static_cast<size_t>(n), 0.0f) {}
    float& operator[](Index i) { return v_[static_cast<size_t>(i)]; }
    const float& operator[](Index i) const { return v_[static_cast<size_t>(i)]; }
    Index size() const { return static_cast<Index>(v_.size()); }
private:
    std::vector<float> v_;
};
class MatrixXf {
public:
    MatrixXf() = default;
    MatrixXf(Index rows, Index cols) : r_(rows), c_(cols), v_(static_cast<size_t>(rows * cols), 0.0f) {}
    float& operator()(Index r, Index c) { return v_[static_cast<size_t>(r * c_ + c)]; }
    const float& operator()(Index r, Index c) const { return v_[static_cast<size_t>(r * c_ + c)]; }
    Index rows() const { return r_; }
    Index cols() const { return c_; }
private:
    Index r_ = 0, c_ = 0;
    std::vector<float> v_;
};
}  // namespace Eigen

namespace matching {
typedef struct BasisChoice_t {
    unsigned short deltaId;
    unsigned short intCoeff;
} BasisChoice;
typedef std::function<math::Matrix(int, const std::vector<BasisChoice>&)> DynamicDictionaryFunction;
typedef std::function<Eigen::MatrixXf(int, const std::vector<BasisChoice>&)> DynamicDictionaryFunctionFast;
int CalcMPDynamic(int K, const double quantization[], std::vector<BasisChoice>& results, const math::Vector& input, const DynamicDictionaryFunction& dynamicDictionary);
int CalcMPDynamicFast(int K, const Eigen::VectorXf& quantization, std::vector<BasisChoice>& results, const Eigen::VectorXf& input, const DynamicDictionaryFunctionFast& dynamicDictionary);
math::Vector FromCoeffsDynamic(int K, const double quantization[], const std::vector<BasisChoice>& coeffs, const DynamicDictionaryFunction& dynamicDictionary);
Eigen::VectorXf FromCoeffsDynamicFast(int K, const Eigen::VectorXf& quantization, const std::vector<BasisChoice>& coeffs, const DynamicDictionaryFunctionFast& dynamicDictionary);
}  // namespace matching

namespace compressed {
void createQuantizationTables(const size_t K, const size_t blockSize, const double bppAllocation, math::Vector& quantY, math::Vector& quantU, math::Vector& quantV);
struct ChannelContext {
    math::Vector Quant;
    std::vector<math::Matrix> DetailBasis;
    matching::DynamicDictionaryFunction Dynamic;
};
struct CompressionContext {
    size_t K{32};
    size_t BlockSize{8};
    math::Matrix BaseDict;
    ChannelContext Y;
    ChannelContext U;
    ChannelContext V;
};
struct ChannelContextFast {
    Eigen::VectorXf Quant;
    std::vector<Eigen::MatrixXf> DetailBasis;
    matching::DynamicDictionaryFunctionFast Dynamic;
};
struct CompressionContextFast {
    size_t K{32};
    size_t BlockSize{8};
    Eigen::MatrixXf BaseDict;
    ChannelContextFast Y;
    ChannelContextFast U;
    ChannelContextFast V;
};
std::unique_ptr<CompressionContext> createCompressionContext(size_t K, size_t blockSize, double bppAllocation);
std::unique_ptr<CompressionContextFast> createCompressionContextFast(size_t K, size_t blockSize, double bppAllocation);
double calculatePSNR(const img::image<img::rgb>* original, const img::image<img::rgb>* decoded);
std::unique_ptr<uint8_t[]> encodeImage(const img::image<img::rgb>* imgIn, const size_t K, const size_t blockSize,
                                       const double quantY[], const double quantU[], const double quantV[],
                                       const matching::DynamicDictionaryFunction& dynamicY, const matching::DynamicDictionaryFunction& dynamicU,
                                       const matching::DynamicDictionaryFunction& dynamicV, size_t& outputByteSize);
std::unique_ptr<uint8_t[]> encodeImageFast(const img::image<img::rgb>* imgIn, const size_t K, const size_t blockSize,
                                           const Eigen::VectorXf& quantY, const Eigen::VectorXf& quantU, const Eigen::VectorXf& quantV,
                                           const matching::DynamicDictionaryFunctionFast& dynamicY, const matching::DynamicDictionaryFunctionFast& dynamicU,
                                           const matching::DynamicDictionaryFunctionFast& dynamicV, size_t& outputByteSize);
std::unique_ptr<img::image<img::rgb>> decodeImage(const uint8_t bytes[], size_t byteSize);
std::unique_ptr<img::image<img::rgb>> decodeImageFast(const uint8_t bytes[], size_t byteSize);
}  // namespace compressed
