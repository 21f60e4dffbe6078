#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/PairWeights.h:34-52
#include <mrpt/containers/yaml.h>
namespace mp2p_icp
{
struct PairWeights
{
    double pt2pt = 1.0, pt2ln = 1.0, pt2pl = 1.0, ln2ln = 1.0, pl2pl = 1.0;
    void   load_from(const mrpt::containers::yaml& p);
};
}  // namespace mp2p_icp
