#pragma once
// stand-in: mrpt/tfest/TMatchingPair.h (36-byte record; Matcher_Points_DistanceThreshold.cpp:106-113)
#include <mrpt/math/types.h>
#include <cstdint>
#include <vector>
namespace mrpt::tfest
{
struct TMatchingPair
{
    uint32_t              globalIdx = 0, localIdx = 0;
    mrpt::math::TPoint3Df global, local;
    float                 errorSquareAfterTransformation = 0;
};
class TMatchingPairList : public std::vector<TMatchingPair>
{
};
}  // namespace mrpt::tfest
