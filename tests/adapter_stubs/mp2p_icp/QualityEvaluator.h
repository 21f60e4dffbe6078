#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/QualityEvaluator.h:31-63
#include <mp2p_icp/Pairings.h>
#include <mp2p_icp/Parameterizable.h>
#include <mp2p_icp/metricmap.h>
#include <mrpt/containers/yaml.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>
namespace mp2p_icp
{
class QualityEvaluator : public mrpt::rtti::CObject, public Parameterizable
{
    DEFINE_VIRTUAL_MRPT_OBJECT(QualityEvaluator, mp2p_icp)
   public:
    struct Result
    {
        double quality      = .0;
        bool   hard_discard = false;
    };
    virtual void   initialize(const mrpt::containers::yaml& params) = 0;
    virtual Result evaluate(const metric_map_t& pcGlobal, const metric_map_t& pcLocal,
                            const mrpt::poses::CPose3D& localPose, const Pairings& pairingsFromICP) const = 0;
};
}  // namespace mp2p_icp
