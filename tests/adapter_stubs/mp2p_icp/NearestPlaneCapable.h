#pragma once
// stand-in: mp2p_icp_map/include/mp2p_icp/NearestPlaneCapable.h:33-52
#include <mp2p_icp/point_plane_pair_t.h>
#include <optional>
namespace mp2p_icp
{
class NearestPlaneCapable
{
   public:
    NearestPlaneCapable() = default;
    virtual ~NearestPlaneCapable();
    struct NearestPlaneResult
    {
        std::optional<point_plane_pair_t> pairing;
        float                             distance = 0;
    };
    virtual NearestPlaneResult nn_search_pt2pl(const mrpt::math::TPoint3Df& point, const float max_search_distance) const = 0;
};
}  // namespace mp2p_icp
