#pragma once
// stand-in: mp2p_icp_map/include/mp2p_icp/plane_patch.h:30-40
#include <mrpt/math/types.h>
namespace mp2p_icp
{
struct plane_patch_t
{
    mrpt::math::TPlane   plane;
    mrpt::math::TPoint3D centroid;
};
}  // namespace mp2p_icp
