#pragma once
#include <vector>
namespace mrpt { template <class T> using aligned_std_vector = std::vector<T>; }
