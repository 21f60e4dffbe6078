#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/Matcher_Points_Base.h:40-129
#include <mp2p_icp/Matcher.h>
#include <mrpt/maps/CPointsMap.h>
namespace mp2p_icp
{
class Matcher_Points_Base : public Matcher
{
   public:
    uint64_t maxLocalPointsPerLayer_ = 0, localPointsSampleSeed_ = 0;
    bool     allowMatchAlreadyMatchedPoints_ = false, allowMatchAlreadyMatchedGlobalPoints_ = false;
    double   bounding_box_intersection_check_epsilon_ = 0.20;
    void     initialize(const mrpt::containers::yaml& params) override;

   protected:
    bool impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const mrpt::poses::CPose3D& localPose,
                    const MatchContext& mc, MatchState& ms, Pairings& out) const override final;

   private:
    virtual void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                                   const mrpt::poses::CPose3D& localPose, MatchState& ms, const layer_name_t& globalName,
                                   const layer_name_t& localName, Pairings& out) const = 0;
};
}  // namespace mp2p_icp
