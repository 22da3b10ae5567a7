// Ordered sum over the slabs of a split reduction (weight gradients of igemm.hip and wino.hip): out[i] = sum_s ws[s][i]
// with a fixed association — bit-reproducible, no atomics.  One launch serves the weight slabs and the bias-gradient
// rows behind them.  Few slabs: a thread owns one float4 and walks the slabs with 8 loads in flight.  Many slabs
// (small weights cut into up to 256 slices): the four waves of a block take contiguous quarters of the slabs for the
// same 64 float4 and meet in LDS, summed in wave order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csg {

struct SlabSeg {
  const float4* ws;   // [nsplit][n4]
  float4* out;        // [n4]
  int64_t n4;
  int blocks;         // blocks serving this segment
};

__device__ __forceinline__ float4 slab_sum(const float4* __restrict__ ws, int64_t n4, int64_t i, int s0, int s1) {
  float4 a = ws[(int64_t)s0 * n4 + i];
  int s = s0 + 1;
  for (; s + 8 <= s1; s += 8) {
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = ws[(int64_t)(s + u) * n4 + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += b[u].x; a.y += b[u].y; a.z += b[u].z; a.w += b[u].w; }
  }
  for (; s < s1; ++s) {
    const float4 b = ws[(int64_t)s * n4 + i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  return a;
}

template <int WIDE>   // WIDE = 1: 64 float4 per block, slabs split over the 4 waves; 0: 256 float4 per block
__global__ __launch_bounds__(256) void k_slab_reduce(SlabSeg a, SlabSeg b, int nsplit) {
  const bool first = (int)blockIdx.x < a.blocks;
  const SlabSeg g = first ? a : b;
  const int blk = first ? (int)blockIdx.x : (int)blockIdx.x - a.blocks;
  if (WIDE) {
    __shared__ float4 part[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blk * 64 + lane;
    const int per = (nsplit + 3) >> 2;
    const int s0 = w * per, s1 = min(nsplit, s0 + per);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool live = i < g.n4 && s0 < s1;
    if (live) acc = slab_sum(g.ws, g.n4, i, s0, s1);
    if (w > 0) part[w - 1][lane] = acc;
    __syncthreads();
    if (w == 0 && i < g.n4) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if ((q + 1) * per < nsplit) {
          const float4 t = part[q][lane];
          acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
        }
      }
      g.out[i] = acc;
    }
  } else {
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    if (i < g.n4) g.out[i] = slab_sum(g.ws, g.n4, i, 0, nsplit);
  }
}

// out_w[n_w] = sum over slabs ws_w[s][n_w]; out_b likewise (n_b may be 0).  Element counts are multiples of 4.
static inline void launch_slab_reduce(const float* ws_w, int64_t n_w, float* out_w, const float* ws_b, int64_t n_b,
                                      float* out_b, int nsplit, hipStream_t s) {
  const bool wide = nsplit >= 16;
  const int per = wide ? 64 : 256;
  SlabSeg a{(const float4*)ws_w, (float4*)out_w, n_w / 4, (int)((n_w / 4 + per - 1) / per)};
  SlabSeg b{(const float4*)ws_b, (float4*)out_b, n_b / 4, (int)((n_b / 4 + per - 1) / per)};
  const unsigned grid = (unsigned)(a.blocks + b.blocks);
  if (wide)
    CSG_LAUNCH(k_slab_reduce<1>, dim3(grid), dim3(256), 0, s, a, b, nsplit);
  else
    CSG_LAUNCH(k_slab_reduce<0>, dim3(grid), dim3(256), 0, s, a, b, nsplit);
}

}  // namespace csg
