#pragma once
// stand-in: mp2p_icp_filters/include/mp2p_icp_filters/GetOrCreatePointLayer.h:31-33
#include <mp2p_icp/metricmap.h>
#include <mrpt/maps/CPointsMap.h>
#include <string>
namespace mp2p_icp_filters
{
[[nodiscard]] mrpt::maps::CPointsMap::Ptr GetOrCreatePointLayer(mp2p_icp::metric_map_t& m, const std::string& layerName,
                                                               bool allowEmptyName = true,
                                                               const std::string& classForLayerCreation = "mrpt::maps::CSimplePointsMap");
}  // namespace mp2p_icp_filters
