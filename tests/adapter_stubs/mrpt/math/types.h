#pragma once
// stand-in: mrpt/math/TPoint3D.h, TPlane.h, TLine3D.h, CMatrixFixed.h
#include <array>
#include <cstddef>
using std::size_t;
namespace mrpt::math
{
template <class T>
struct TPoint3D_
{
    T x = 0, y = 0, z = 0;
    TPoint3D_() = default;
    TPoint3D_(T X, T Y, T Z) : x(X), y(Y), z(Z) {}
    T&       operator[](size_t i);
    const T& operator[](size_t i) const;
};
using TPoint3D  = TPoint3D_<double>;
using TPoint3Df = TPoint3D_<float>;
using TVector3D = TPoint3D;
struct TPlane
{
    std::array<double, 4> coefs{{0, 0, 0, 0}};
};
using TPlane3D = TPlane;
struct TLine3D
{
    TPoint3D  pBase;
    TVector3D director;
};
template <class T, int R, int C>
struct CMatrixFixed
{
    T&       operator()(int r, int c);
    const T& operator()(int r, int c) const;
};
using CMatrixDouble33 = CMatrixFixed<double, 3, 3>;
using CMatrixDouble66 = CMatrixFixed<double, 6, 6>;
}  // namespace mrpt::math
