#pragma once
// stand-in: mrpt/math/distributions.h (confidenceIntervalsFromHistogram, called at Matcher_Adaptive.cpp:203-205)
#include <vector>
namespace mrpt::math
{
template <class CONTAINER>
void confidenceIntervalsFromHistogram(const CONTAINER& x, const CONTAINER& p, double& out_lower_conf_interval,
                                      double& out_upper_conf_interval, const double confidenceInterval = 0.1);
}  // namespace mrpt::math
