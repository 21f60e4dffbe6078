#pragma once
// stand-in: mp2p_icp_map/include/mp2p_icp/metricmap.h (:91 layers, :273-295 MapTo*)
#include <mp2p_icp/NearestPlaneCapable.h>
#include <mp2p_icp/layer_name_t.h>
#include <mrpt/maps/CPointsMap.h>
#include <map>
#include <mrpt/core/exceptions.h>
#include <vector>
namespace mrpt::maps
{
class NearestNeighborsCapable
{
   public:
    size_t nn_index_count() const;
    bool   nn_has_indices_or_ids() const;
};
}  // namespace mrpt::maps
namespace mp2p_icp
{
class metric_map_t
{
   public:
    std::map<layer_name_t, mrpt::maps::CMetricMap::Ptr> layers;
    std::vector<int> lines, planes;  // (element types irrelevant here: only size() is used)
};
const mrpt::maps::NearestNeighborsCapable* MapToNN(const mrpt::maps::CMetricMap& map, bool throwIfNotImplemented = false);
const mrpt::maps::CPointsMap* MapToPointsMap(const mrpt::maps::CMetricMap& map);
const mp2p_icp::NearestPlaneCapable* MapToNP(const mrpt::maps::CMetricMap& map, bool throwIfNotImplemented = false);
}  // namespace mp2p_icp
