#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/Matcher.h:30-108
#include <mp2p_icp/Pairings.h>
#include <mp2p_icp/Parameterizable.h>
#include <mp2p_icp/metricmap.h>
#include <mp2p_icp/pointcloud_bitfield.h>
#include <mrpt/core/exceptions.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>
namespace mp2p_icp
{
struct MatchContext
{
    uint32_t icpIteration = 0;
};
struct MatchState
{
    MatchState() = default;
    MatchState(const metric_map_t& pcGlobal, const metric_map_t& pcLocal);  // Matcher.h:46-50
    pointcloud_bitfield_t localPairedBitField, globalPairedBitField;
};
class Matcher : public mrpt::rtti::CObject, public Parameterizable
{
    DEFINE_VIRTUAL_MRPT_OBJECT(Matcher, mp2p_icp)
   public:
    virtual void initialize(const mrpt::containers::yaml& params);
    virtual bool match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const mrpt::poses::CPose3D& localPose,
                       const MatchContext& mc, MatchState& ms, Pairings& out) const;
    uint32_t runFromIteration = 0, runUpToIteration = 0;
    bool     enabled = true;

   protected:
    virtual bool impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal,
                            const mrpt::poses::CPose3D& localPose, const MatchContext& mc, MatchState& ms,
                            Pairings& out) const = 0;
};
}  // namespace mp2p_icp
