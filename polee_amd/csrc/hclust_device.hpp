// Device-side tree construction: the rounds variant of the clustering heuristic (hclust.cpp, polee_hclust_parallel) as kernels.
#pragma once
#include "common.hpp"

namespace polee {
}  // namespace polee
