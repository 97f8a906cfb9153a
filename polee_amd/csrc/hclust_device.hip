// Device-side tree construction (the rounds variant of hclust.cpp as kernels): see hclust_device.hpp.
#include <cstring>
#include <cstdio>
#include <hip/hip_runtime.h>

#include "hclust_device.hpp"

namespace polee {
}  // namespace polee
