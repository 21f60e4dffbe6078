#pragma once
// stand-in: mp2p_icp_filters/include/mp2p_icp_filters/FilterDecimateVoxels.h:36-47 (enum), :73-83 (class, declarations only)
#include <mp2p_icp_filters/FilterBase.h>
#include <cstdint>
namespace mp2p_icp_filters
{
enum class DecimateMethod : uint8_t
{
    FirstPoint = 0,
    ClosestToAverage,
    VoxelAverage,
    RandomPoint
};
class FilterDecimateVoxels : public FilterBase
{
    DEFINE_MRPT_OBJECT(FilterDecimateVoxels, mp2p_icp_filters)
   public:
    FilterDecimateVoxels();
    void initialize(const mrpt::containers::yaml& c) override;
    void filter(mp2p_icp::metric_map_t& inOut) const override;
};
}  // namespace mp2p_icp_filters
