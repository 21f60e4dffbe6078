#pragma once
// stand-in: mrpt/maps/CMetricMap.h
#include <mrpt/serialization/CSerializable.h>
namespace mrpt::maps
{
class CMetricMap : public mrpt::serialization::CSerializable
{
   public:
    using Ptr = std::shared_ptr<CMetricMap>;
    virtual bool isEmpty() const = 0;
};
}  // namespace mrpt::maps
