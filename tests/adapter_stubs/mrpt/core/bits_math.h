#pragma once
namespace mrpt { template <class T> inline T square(const T x) { return x * x; } }
