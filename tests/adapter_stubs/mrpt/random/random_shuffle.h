#pragma once
// stand-in: mrpt/random/random_shuffle.h
#include <cstddef>
namespace mrpt::random
{
template <class RandomIt, class URBG>
void partial_shuffle(RandomIt first, RandomIt last, URBG&& g, const std::size_t N);
}
