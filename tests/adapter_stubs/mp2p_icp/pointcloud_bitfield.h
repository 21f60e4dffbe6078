#pragma once
// stand-in: mp2p_icp_map/include/mp2p_icp/pointcloud_bitfield.h:46-92 -- the same public interface and
// the same private members (the plugin reaches `dense_` through an explicit template instantiation)
#include <mp2p_icp/layer_name_t.h>
#include <cstdint>
#include <map>
#include <optional>
#include <set>
#include <vector>
namespace mp2p_icp
{
struct pointcloud_bitfield_t
{
    struct DenseOrSparseBitField
    {
       public:
        void assign(size_t numElements, bool dense);
        bool operator[](const size_t id) const;
        void mark_as_set(const size_t id);

       private:
        std::optional<std::vector<bool>> dense_;
        std::set<uint64_t>               sparse_;
    };
    std::map<layer_name_t, DenseOrSparseBitField> point_layers;
    std::vector<bool>                             lines, planes;
};
}  // namespace mp2p_icp
