#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/Solver.h:43-104, OptimalTF_Result.h:32-39
#include <mp2p_icp/Pairings.h>
#include <mp2p_icp/Parameterizable.h>
#include <mrpt/core/exceptions.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>
#include <optional>
namespace mp2p_icp
{
struct OptimalTF_Result
{
    mrpt::poses::CPose3D optimalPose;
    double               optimalScale = 1.0;
    OutlierIndices       outliers;
};
struct SolverContext
{
    std::optional<mrpt::poses::CPose3D>               guessRelativePose;
    std::optional<mrpt::poses::CPose3DPDFGaussianInf> prior;
    std::optional<uint32_t>                           icpIteration;
};
class Solver : public mrpt::rtti::CObject, public Parameterizable
{
    DEFINE_VIRTUAL_MRPT_OBJECT(Solver, mp2p_icp)
   public:
    virtual void initialize(const mrpt::containers::yaml& params);

   protected:
    virtual bool impl_optimal_pose(const Pairings& pairings, OptimalTF_Result& out, const SolverContext& sc) const = 0;
};
}  // namespace mp2p_icp
