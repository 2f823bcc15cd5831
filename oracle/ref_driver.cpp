// ref_driver.cpp -- ORACLE (test infrastructure, not product code).
//
// Thin extern "C" driver over the REFERENCE's own SimpleMatrix and
// ImageHelper/misc sources, which oracle/Makefile compiles in place from
// /root/reference (no copies, no stand-in headers).  Built into
// oracle/_ref/libref.so and used only by tests/ to check the plain-C
// restatement (mpo_*.c) bit-for-bit against the reference's object code:
//   math::SymmetricEigenDecomposition   SimpleMatrix/src/symmeigen.cpp:34
//   math::Multiply(Matrix, Vector)      SimpleMatrix/src/mathmatrix.cpp:426
//   math::Vector::Scale / Subtract      SimpleMatrix/src/mathvector.cpp:140 / :116
//   img::YUVFromRGB / RGBFromYUV        ImageHelper/src/misc.cpp:7-36
#include "/root/reference/SimpleMatrix/inc/mathmatrix.h"
#include "/root/reference/SimpleMatrix/inc/mathvector.h"
#include "/root/reference/SimpleMatrix/inc/symmeigen.h"
#include "/root/reference/ImageHelper/inc/misc.h"
#include <cstdint>
#include <cstring>

extern "C" {

void ref_symm_eigen(const double* a, int n, double* vec, double* val)
{
    math::Matrix m(static_cast<size_t>(n), static_cast<size_t>(n), a);
    math::SymmetricEigenDecomposition deco(m);
    math::Matrix v = deco.EigenVectors();
    math::Vector e = deco.EigenValues();
    std::memcpy(vec, v.Data(), sizeof(double) * n * n);
    std::memcpy(val, e.Data(), sizeof(double) * n);
}

void ref_multiply(const double* mat, int rows, int cols, const double* vec, double* out)
{
    math::Matrix m(static_cast<size_t>(rows), static_cast<size_t>(cols), mat);
    math::Vector v(static_cast<size_t>(cols), vec);
    math::Vector p = math::Multiply(m, v);
    std::memcpy(out, p.Data(), sizeof(double) * rows);
}

// newEntry.Scale(coeff); residual.Subtract(newEntry);   (MatchingPursuit.cpp:70-71)
void ref_scale_subtract(double* residual, const double* atom, double coeff, int n)
{
    math::Vector r(static_cast<size_t>(n), residual);
    math::Vector a(static_cast<size_t>(n), atom);
    a.Scale(coeff);
    r.Subtract(a);
    std::memcpy(residual, r.Data(), sizeof(double) * n);
}

void ref_yuv_from_rgb(uint8_t r, uint8_t g, uint8_t b, double* yuv)
{
    img::yuv c = img::YUVFromRGB(img::rgb{r, g, b});
    yuv[0] = c.y; yuv[1] = c.u; yuv[2] = c.v;
}

void ref_rgb_from_yuv(double y, double u, double v, uint8_t* rgb)
{
    img::rgb c = img::RGBFromYUV(img::yuv{y, u, v});
    rgb[0] = c.r; rgb[1] = c.g; rgb[2] = c.b;
}

}
