#pragma once
// stand-in: mp2p_icp_map/include/mp2p_icp/point_plane_pair_t.h:34-46
#include <mp2p_icp/plane_patch.h>
#include <vector>
namespace mp2p_icp
{
struct point_plane_pair_t
{
    plane_patch_t         pl_global;
    mrpt::math::TPoint3Df pt_local;
};
using MatchedPointPlaneList = std::vector<point_plane_pair_t>;
}  // namespace mp2p_icp
