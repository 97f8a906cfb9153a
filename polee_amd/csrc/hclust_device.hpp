// Device-side tree construction: the rounds variant of the clustering heuristic (hclust.cpp, polee_hclust_parallel) as kernels.
#pragma once
#include <string>

#include "common.hpp"

namespace polee {

// X by columns (CSC, 1-based, host arrays) -> the serialised tree, the same arrays as hclust_build_rounds gives
polee_status hclust_rounds_device(polee_ctx *ctx, int64_t m, int64_t n, const void *colptr, int colptr_bytes, const uint32_t *rowval,
                                  int32_t *node_parent_idxs, int32_t *node_js);

// the same from Xt (the rows of X: tcolptr u64 [m + 1] and trowval u32, 1-based) in DEVICE memory -- an xbuild result
polee_status hclust_rounds_device_from_xt(polee_ctx *ctx, int64_t m, int64_t n, const uint64_t *d_tcolptr, const uint32_t *d_trowval,
                                          int32_t *node_parent_idxs, int32_t *node_js);

// hclust.cpp: the stages every variant ends with (components without a common read joined smallest first, order_nodes) from
// plain arrays: nodes 1..n are the leaves (leaf_transcript[q] = 0-based transcript of node q + 1), nodes n+1 .. num_nodes-1 the
// merges (left / right), alive / set_len per node
std::string hclust_finish_from_arrays(int64_t n, uint32_t num_nodes, const int32_t *left, const int32_t *right, const uint8_t *alive,
                                      const uint32_t *set_len, const uint32_t *leaf_transcript, int32_t *node_parent_idxs, int32_t *node_js);

}  // namespace polee
