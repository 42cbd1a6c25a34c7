// Round 6 — BatchNorm-backward sums and ReLU masks taken in the epilogue of the kernel that PRODUCES the gradient.
//
// Train-mode BatchNorm backward (map_encoder.py:19-29,94-112, mg_map_policy.py:78-100 of the reference: 13 of them per update) needs
// two per-channel sums over the whole batch before it can write anything:  dbeta = sum g,  dgamma = sum g * xhat,  with
// g = dy * (relu output > 0) and xhat = (x - mean) * invstd.  Rounds 1-5 took them in a pass of their own over dy and x
// (col_reduce_kernel<2>: 18 launches, 1.6 GB, 0.39 ms per update).  Here the kernel that writes dy — a backward-data convolution,
// the three-way gradient add, the upsampling's backward — reads the matching piece of x while it holds the gradient piece in
// registers, stores the MASKED gradient, and every workgroup stores its partial sums as one block of `part` [blocks][2][C]
// float64: the layout col_reduce_kernel wrote, so the same finalize kernel adds the blocks in block order.  Plain stores, every
// element written exactly once: bit-reproducible.  (First form of this round: float64 atomics into 8 slabs that the apply pass
// reduced in its prologue — 16.2 ms per update instead of 10.5: ~80 000 atomics per 128-byte line at ~40 ns each.)
// Same values, same expressions as the pass it replaces (the mask is recomputed from x exactly as bn_apply8_kernel computed the
// activation); the sums differ in summation order only.
//
// The same hook carries the plain ReLU mask of a convolution whose ReLU was fused into its forward epilogue (mode 1): the
// backward-data kernel of the NEXT layer masks its output with that layer's saved input (= this layer's ReLU output).
#pragma once
#include "wsmg_common.h"

typedef unsigned int u32x4a __attribute__((ext_vector_type(4)));

// device-side view of wsmg_bn_aux_t (include/wsmgmap.h); mode 0 = nothing
struct BnAux {
  const bf16_t* z;       // mode 1: the ReLU output the gradient is masked with; modes 2 / 3: the BatchNorm's INPUT x
  const float* mean;     // modes 2 / 3: [C] batch statistics saved by the forward pass
  const float* invstd;
  const float* gamma;    // mode 2: [C] affine parameters (the ReLU mask is recomputed as (x - mean) invstd gamma + beta > 0)
  const float* beta;
  double* part;          // modes 2 / 3: [blocks][2][C] float64: block b = the partial sums (sum g, sum g xhat) of workgroup / tile b
  int mode, c0, C, ld;   // gradient channels [c0, c0 + C) <-> z channels [0, C); ld = z's pixel pitch in elements
};

static inline BnAux bn_aux_host(const wsmg_bn_aux_t* p) {
  BnAux a{};
  if (!p) return a;
  a.z = (const bf16_t*)p->z; a.mean = p->mean; a.invstd = p->invstd; a.gamma = p->gamma; a.beta = p->beta; a.part = p->part;
  a.mode = p->mode; a.c0 = p->c0; a.C = p->C; a.ld = p->ld;
  return a;
}
// arguments a launcher refuses: WSMG_EINVAL
static inline bool bn_aux_bad(const wsmg_bn_aux_t* p, int out_channels) {
  if (!p || p->mode == 0) return false;
  if (p->mode < 0 || p->mode > 3 || !p->z || (p->C & 7) || (p->c0 & 7) || p->C <= 0 || p->c0 < 0 || p->c0 + p->C > out_channels) return true;
  if (p->ld < p->C || (p->ld & 7) || ((uintptr_t)p->z & 15)) return true;
  if (p->mode >= 2 && (!p->mean || !p->invstd || !p->part || p->cap_blocks <= 0)) return true;
  if (p->mode == 2 && (!p->gamma || !p->beta)) return true;
  return false;
}
// the launcher's report of how many blocks its launch fills (WSMG_ENOMEM if `part` is too small)
static inline int bn_aux_blocks(wsmg_bn_aux_t* p, int64_t blocks) {
  if (!p || p->mode < 2) return 0;
  if (blocks > p->cap_blocks) return WSMG_ENOMEM;
  p->blocks = (int)blocks;
  return 0;
}

// per-thread state of one 8-channel group: the channel constants and the two running sums (S = float where a thread adds at most
// a tile's 4-8 rows before the sums go on in float64 — the convolution epilogues, whose registers are scarce —, else double)
template <class S>
struct BnAuxAccT {
  f32x4 mu[2], is[2], gm[2], bt[2];
  S s0[8], s1[8];
  bool on;               // this thread's channel group lies inside [c0, c0 + C)
  int cz;                // first z channel of the group
};
typedef BnAuxAccT<double> BnAuxAcc;
typedef BnAuxAccT<float> BnAuxAccF;

// n: first gradient channel of the thread's 8-channel group (fixed for the whole launch / tile)
template <class S>
__device__ __forceinline__ void bn_aux_begin(const BnAux& a, int n, BnAuxAccT<S>& t) {
  t.on = a.mode != 0 && n >= a.c0 && n < a.c0 + a.C;
  t.cz = n - a.c0;
#pragma unroll
  for (int j = 0; j < 8; ++j) { t.s0[j] = (S)0; t.s1[j] = (S)0; }
  if (t.on && a.mode >= 2) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      t.mu[k] = *reinterpret_cast<const f32x4*>(a.mean + t.cz + 4 * k);
      t.is[k] = *reinterpret_cast<const f32x4*>(a.invstd + t.cz + 4 * k);
      if (a.mode == 2) {
        t.gm[k] = *reinterpret_cast<const f32x4*>(a.gamma + t.cz + 4 * k);
        t.bt[k] = *reinterpret_cast<const f32x4*>(a.beta + t.cz + 4 * k);
      }
    }
  }
}

template <class S>
__device__ __forceinline__ u32x4a bn_aux_load(const BnAux& a, const BnAuxAccT<S>& t, size_t pixel) {
  return *reinterpret_cast<const u32x4a*>(a.z + pixel * (size_t)a.ld + t.cz);
}

// g: 8 bf16 gradient values of one pixel (as they will be stored); zr: the matching piece of z.  Returns the piece to store.
template <class S>
__device__ __forceinline__ u32x4a bn_aux_apply(const BnAux& a, BnAuxAccT<S>& t, const u32x4a g, const u32x4a zr) {
  u32x4a out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float z0 = __uint_as_float(zr[q] << 16), z1 = __uint_as_float(zr[q] & 0xffff0000u);
    const int k = q >> 1, j0 = (2 * q) & 3, j1 = j0 + 1;
    bool keep0 = true, keep1 = true;
    if (a.mode == 1) {
      keep0 = z0 > 0.f;
      keep1 = z1 > 0.f;
    } else if (a.mode == 2) {   // the forward's expression (bn_apply8_kernel): (x - mean) * invstd * gamma + beta
      keep0 = ((z0 - t.mu[k][j0]) * t.is[k][j0] * t.gm[k][j0] + t.bt[k][j0]) > 0.f;
      keep1 = ((z1 - t.mu[k][j1]) * t.is[k][j1] * t.gm[k][j1] + t.bt[k][j1]) > 0.f;
    }
    // (the masked values are the gradient's own bits or +0: no re-rounding)
    const unsigned o = (keep0 ? (g[q] & 0xffffu) : 0u) | (keep1 ? (g[q] & 0xffff0000u) : 0u);
    out[q] = o;
    if (a.mode >= 2) {
      const float g0 = __uint_as_float(o << 16), g1 = __uint_as_float(o & 0xffff0000u);
      const float h0 = (z0 - t.mu[k][j0]) * t.is[k][j0], h1 = (z1 - t.mu[k][j1]) * t.is[k][j1];
      t.s0[2 * q] += (S)g0;
      t.s1[2 * q] += (S)g0 * (S)h0;
      t.s0[2 * q + 1] += (S)g1;
      t.s1[2 * q + 1] += (S)g1 * (S)h1;
    }
  }
  return out;
}

// End of a tile / launch, ALL threads of the workgroup (NW waves): the sums of the threads that share a channel group — lanes
// congruent modulo G inside a wave (G a power of two <= 64, thread -> group = tid % G) — are added up by shuffles, then over the
// waves in wave order through `lds` (NW x 16 x G doubles, free for this), and stored as block `blk` of `part`.  Groups outside the
// hook's channel range store nothing.  zero: store zeros (a workgroup of the grid that has no pixels).
template <int NW, class S>
__device__ __forceinline__ void bn_aux_store_block(const BnAux& a, BnAuxAccT<S>& t, int G, double* lds, int blk) {
  if (a.mode < 2) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double u = t.on ? (double)t.s0[j] : 0.0, v = t.on ? (double)t.s1[j] : 0.0;
    for (int off = G; off < 64; off <<= 1) {
      u += __shfl_xor(u, off, 64);
      v += __shfl_xor(v, off, 64);
    }
    if (lane < G) { lds[(wave * 16 + j) * G + lane] = u; lds[(wave * 16 + 8 + j) * G + lane] = v; }
  }
  __syncthreads();
  if (wave == 0 && lane < G && t.on) {
    double* const st = a.part + (size_t)blk * 2 * a.C + t.cz;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double u = 0.0, v = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) { u += lds[(w * 16 + j) * G + lane]; v += lds[(w * 16 + 8 + j) * G + lane]; }
      st[j] = u;
      st[a.C + j] = v;
    }
  }
  __syncthreads();
}
