#pragma once
// stand-in: mrpt/maps/CPointsMap.h (the buffers the matchers read, Matcher_Points_Base.cpp:201-203)
#include <mrpt/core/aligned_std_vector.h>
#include <mrpt/maps/CMetricMap.h>
namespace mrpt::maps
{
class CPointsMap : public CMetricMap
{
   public:
    using Ptr = std::shared_ptr<CPointsMap>;
    size_t size() const;
    bool   empty() const;
    bool   isEmpty() const override;
    const mrpt::aligned_std_vector<float>& getPointsBufferRef_x() const;
    const mrpt::aligned_std_vector<float>& getPointsBufferRef_y() const;
    const mrpt::aligned_std_vector<float>& getPointsBufferRef_z() const;
    // FilterDecimateVoxels.cpp:145-189, 236-244
    void reserve(size_t n);
    void insertPointFast(float x, float y, float z);
    void insertPointFrom(const CPointsMap& source, size_t sourceIndex);
    void mark_as_modified() const;
};
}  // namespace mrpt::maps
