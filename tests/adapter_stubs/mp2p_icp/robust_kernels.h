#pragma once
// stand-in: mp2p_icp/include/mp2p_icp/robust_kernels.h:33-43
#include <cstdint>
namespace mp2p_icp
{
enum class RobustKernel : uint8_t
{
    None = 0,
    GemanMcClure,
    Cauchy,
};
}
