#pragma once
// stand-in: mp2p_icp_filters/include/mp2p_icp_filters/FilterBase.h:40-71 (declarations only)
#include <mp2p_icp/Parameterizable.h>
#include <mp2p_icp/metricmap.h>
#include <mrpt/containers/yaml.h>
#include <mrpt/rtti/CObject.h>
namespace mp2p_icp_filters
{
class FilterBase : public mrpt::rtti::CObject, public mp2p_icp::Parameterizable
{
    DEFINE_VIRTUAL_MRPT_OBJECT(FilterBase, mp2p_icp_filters)
   public:
    FilterBase();
    virtual ~FilterBase();
    virtual void initialize(const mrpt::containers::yaml& cfg_block) = 0;
    virtual void filter(mp2p_icp::metric_map_t& inOut) const        = 0;
};
}  // namespace mp2p_icp_filters
