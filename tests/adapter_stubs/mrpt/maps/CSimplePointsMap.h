#pragma once
// stand-in: mrpt/maps/CSimplePointsMap.h
#include <mrpt/maps/CPointsMap.h>
namespace mrpt::maps
{
class CSimplePointsMap : public CPointsMap
{
   public:
    static const mrpt::rtti::TRuntimeClassId runtimeClassId;

   protected:
    uint8_t serializeGetVersion() const override;
    void    serializeTo(mrpt::serialization::CArchive& out) const override;
    void    serializeFrom(mrpt::serialization::CArchive& in, uint8_t version) override;
};
}  // namespace mrpt::maps
