#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/WeightParameters.h:34-72
#include <mp2p_icp/PairWeights.h>
#include <mp2p_icp/robust_kernels.h>
#include <mrpt/poses/CPose3D.h>
#include <optional>
namespace mp2p_icp
{
struct WeightParameters
{
    bool                                use_scale_outlier_detector = false;
    double                              scale_outlier_threshold    = 1.20;
    PairWeights                         pair_weights;
    RobustKernel                        robust_kernel = RobustKernel::None;
    std::optional<mrpt::poses::CPose3D> currentEstimateForRobust;
    double                              robust_kernel_param = 1.0;
    void                                load_from(const mrpt::containers::yaml& p);
};
}  // namespace mp2p_icp
