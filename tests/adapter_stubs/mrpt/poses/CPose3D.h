#pragma once
// stand-in: mrpt/poses/CPose3D.h, CPose3DPDFGaussianInf.h
#include <mrpt/math/types.h>
namespace mrpt::poses
{
class CPose3D
{
   public:
    CPose3D() = default;
    CPose3D(const mrpt::math::CMatrixDouble33& rot, const mrpt::math::TPoint3D& xyz);
    const mrpt::math::CMatrixDouble33& getRotationMatrix() const;
    double x() const;
    double y() const;
    double z() const;
};
class CPose3DPDFGaussianInf
{
   public:
    CPose3D                     mean;
    mrpt::math::CMatrixDouble66 cov_inv;
};
}  // namespace mrpt::poses
