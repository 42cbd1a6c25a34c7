// Round 6 — the ReLU mask of a fused-ReLU convolution applied in the epilogue of the kernel that PRODUCES its output's gradient.
//
// map_encoded_linear / map_classified_linear (mg_map_policy.py:89-96 of the reference: Conv2d + ReLU, no BatchNorm) have their ReLU fused
// into the forward epilogue and write straight into their channel slices of the tensor map_cated_linear reads (torch.cat at :197).
// Backward, rounds 1-5 ran a mask pass per producer over its slice of the concatenation's gradient (relu_bwd8_rows_kernel: read dy and
// y, write the masked copy).  Here map_cated_linear's backward-data kernel masks the 16-byte gradient piece it is about to store with the
// matching piece of ITS OWN saved input — which is exactly the two ReLU outputs — and stores it as the two producers' contiguous parts.
//
// (The same hook carried BatchNorm-backward sums for one round-6 experiment — sum g and sum g xhat taken here instead of in the
//  reduction pass — built twice, slower twice, removed: profiles/r06_bn_producer_sums_negative.txt.)
#pragma once
#include "wsmg_common.h"

typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));

// g: 8 bf16 gradient values of one pixel as they will be stored; zr: the 8 ReLU outputs of the same pixel and channels.
// Returns g where z > 0, +0 elsewhere (the gradient's own bits: no re-rounding).
__device__ __forceinline__ u32x4a relu_mask8(const u32x4a g, const u32x4a zr) {
  u32x4a out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = __uint_as_float(zr[q] << 16), z1 = __uint_as_float(zr[q] & 0xffff0000u);
    out[q] = (z0 > 0.f ? (g[q] & 0xffffu) : 0u) | (z1 > 0.f ? (g[q] & 0xffff0000u) : 0u);
  }
  return out;
}
