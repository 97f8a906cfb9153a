#include <hip/hip_runtime.h>
__global__ void k(int *p) { if (p) *p = 1; }
