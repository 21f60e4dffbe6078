#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/Pairings.h:37-169
#include <mp2p_icp/point_plane_pair_t.h>
#include <mrpt/tfest/TMatchingPair.h>
#include <utility>
#include <vector>
namespace mp2p_icp
{
struct matched_plane_t
{
    plane_patch_t p_global, p_local;
};
struct matched_line_t
{
    mrpt::math::TLine3D ln_global, ln_local;
};
struct point_line_pair_t
{
    mrpt::math::TLine3D  ln_global;
    mrpt::math::TPoint3D pt_local;
};
struct Pairings
{
    virtual ~Pairings();
    mrpt::tfest::TMatchingPairList              paired_pt2pt;
    std::vector<point_line_pair_t>              paired_pt2ln;
    MatchedPointPlaneList                       paired_pt2pl;
    std::vector<matched_line_t>                 paired_ln2ln;
    std::vector<matched_plane_t>                paired_pl2pl;
    uint64_t                                    potential_pairings = 0;
    virtual size_t                              size() const;   // Pairings.h:141, Pairings.cpp:143-147
    std::vector<std::pair<std::size_t, double>> point_weights;
};
struct OutlierIndices
{
    std::vector<std::size_t> point2point, line2line, plane2plane;
};
}  // namespace mp2p_icp
