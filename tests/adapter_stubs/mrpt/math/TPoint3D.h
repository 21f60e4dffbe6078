#pragma once
#include <mrpt/math/types.h>
