// K9/K11 — normalisation statistics, SPADE modulation, LeakyReLU, and the small NHWC
// resampling kernels (K7, avg-pool).  All HBM-bound streaming kernels: 16-byte accesses per lane,
// channel-fastest NHWC, deterministic two-stage reductions (fp32 partials per pixel chunk,
// combined in fp64) so that results are bit-reproducible and SyncBN can all-reduce one small
// message per norm.
//
// Reference: sync_batchnorm/batchnorm.py:63-93,128-145 (statistics), normalization.py:96-110
// (SPADE), architecture.py:53-54,67-68 (LeakyReLU), normalization.py:44 +
// discriminator.py:181-185 (InstanceNorm + LeakyReLU), generator.py:48 (nearest 2x),
// discriminator.py:92-93 (avg_pool2d 3/2/1 without pad count).
#include "csg_common.h"

using namespace csg;

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }
__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float lrelu_g(float pre, float s) { return pre > 0.f ? 1.f : s; }

// --------------------------------------------------------------------------------- statistics
// grid (nchunk, G); partial[(g*nchunk + chunk)*2C + {c | C + c}] = {sum x, sum x^2} over the chunk.
// Sums are carried in fp64 from the first addition on: var = E[x^2] - E[x]^2 cancels mean^2 / var leading digits,
// and the discriminator's InstanceNorm maps (mean^2 >> var on 16 641-pixel planes) lost 3-4 digits of inv_std
// with fp32 partial sums (profiles/archive/r02_band_*.txt: 1e-4 gradient error where the fp32 reference has 1e-6).
// x and x*x are exact in fp64, so the result is the correctly rounded statistic whatever the chunking.
struct d4 {
  double x, y, z, w;
};
__device__ __forceinline__ d4 d4zero() { return d4{0.0, 0.0, 0.0, 0.0}; }

__global__ __launch_bounds__(256) void k_norm_stats_partial(const float* __restrict__ x, int64_t P, int C,
                                                             int64_t x_cs, int nchunk, double* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) double smd[];  // [R][Qc][8]
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, g = blockIdx.y;
  const int64_t per = (P + nchunk - 1) / nchunk;
  const int64_t p0 = chunk * per, p1 = min(P, p0 + per);
  const int Q = C >> 2;
  double* out = partial + ((int64_t)g * nchunk + chunk) * 2 * C;
  for (int qb = 0; qb < Q; qb += 256) {
    const int Qc = min(256, Q - qb);
    const int R = 256 / Qc;
    const int q = tid % Qc, rr = tid / Qc;
    d4 s = d4zero(), ss = d4zero();
    if (rr < R) {
      const float* xp = x + ((int64_t)g * P) * x_cs + (qb + q) * 4;
      // four loads in flight per thread (one per iteration leaves HBM a third busy); same order of additions
      int64_t p = p0 + rr;
      for (; p + 3 * R < p1; p += 4 * R) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(xp + (p + u * R) * x_cs);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double a = v[u].x, b = v[u].y, c = v[u].z, d = v[u].w;
          s.x += a; s.y += b; s.z += c; s.w += d;
          ss.x += a * a; ss.y += b * b; ss.z += c * c; ss.w += d * d;
        }
      }
      for (; p < p1; p += R) {
        const float4 v = ld4(xp + p * x_cs);
        const double a = v.x, b = v.y, c = v.z, d = v.w;
        s.x += a; s.y += b; s.z += c; s.w += d;
        ss.x += a * a; ss.y += b * b; ss.z += c * c; ss.w += d * d;
      }
      double* o = &smd[(rr * Qc + q) * 8];
      o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
      o[4] = ss.x; o[5] = ss.y; o[6] = ss.z; o[7] = ss.w;
    }
    __syncthreads();
    if (tid < Qc) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += smd[(r * Qc + tid) * 8 + e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        out[(qb + tid) * 4 + e] = a[e];
        out[C + (qb + tid) * 4 + e] = a[4 + e];
      }
    }
    __syncthreads();
  }
}

// Second stage of every two-stage reduction: out[g][j] = sum over chunks of partial[g][chunk][j],
// accumulated in fp64 in a fixed order.  Block = 32 columns x 8 chunk lanes, 4 loads in flight.
template <typename OutT>
__global__ __launch_bounds__(256) void k_partial_reduce(const double* __restrict__ partial, int ncols, int nchunk,
                                                         int out_cols, OutT* __restrict__ out) {
  __shared__ double sm[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cl, g = blockIdx.y;
  double a = 0.0;
  if (j < out_cols) {
    const double* p = partial + (int64_t)g * nchunk * ncols + j;
    int c = rl;
    for (; c + 24 < nchunk; c += 32) {
      const double v0 = p[(int64_t)c * ncols], v1 = p[(int64_t)(c + 8) * ncols];
      const double v2 = p[(int64_t)(c + 16) * ncols], v3 = p[(int64_t)(c + 24) * ncols];
      a += v0;
      a += v1;
      a += v2;
      a += v3;
    }
    for (; c < nchunk; c += 8) a += p[(int64_t)c * ncols];
  }
  sm[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && j < out_cols) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sm[r][cl];
    out[(int64_t)g * out_cols + j] = (OutT)t;
  }
}

// One-rank statistics: the second stage of the reduction and the finalisation in ONE launch.  A block owns 16 channels:
// columns c0..c0+15 (sums) and C+c0..C+c0+15 (sums of squares) of the partial table, each reduced exactly as
// k_partial_reduce reduces a column (same chunk lanes, same order: bit-identical sums), then mean / invstd / running
// statistics as k_norm_finalize forms them (mode 0).  Up to two modules' running buffers (norm_0 and norm_s of a residual
// block see the same batch).
__global__ __launch_bounds__(256) void k_partial_reduce_finalize(const double* __restrict__ partial, int C, int nchunk,
                                                                  double count, float eps, float* __restrict__ mean,
                                                                  float* __restrict__ invstd, float* __restrict__ rmean,
                                                                  float* __restrict__ rvar, float* __restrict__ rmean2,
                                                                  float* __restrict__ rvar2, float momentum) {
  __shared__ double sm[8][32];
  __shared__ double tot[32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int g = blockIdx.y, c = blockIdx.x * 16 + (cl & 15);
  const int ncols = 2 * C;
  const int j = (cl < 16 ? 0 : C) + c;
  double a = 0.0;
  if (c < C) {
    const double* p = partial + (int64_t)g * nchunk * ncols + j;
    int k = rl;
    for (; k + 24 < nchunk; k += 32) {
      const double v0 = p[(int64_t)k * ncols], v1 = p[(int64_t)(k + 8) * ncols];
      const double v2 = p[(int64_t)(k + 16) * ncols], v3 = p[(int64_t)(k + 24) * ncols];
      a += v0;
      a += v1;
      a += v2;
      a += v3;
    }
    for (; k < nchunk; k += 8) a += p[(int64_t)k * ncols];
  }
  sm[rl][cl] = a;
  __syncthreads();
  if (rl == 0) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += sm[r][cl];
    tot[cl] = t;
  }
  __syncthreads();
  if (threadIdx.x < 16 && c < C) {
    const double S = tot[threadIdx.x], SS = tot[threadIdx.x + 16];
    const double m = S / count;
    double var = (SS - S * m) / count;
    if (var < 0.0) var = 0.0;
    const float fv = (float)var;
    mean[(int64_t)g * C + c] = (float)m;
    invstd[(int64_t)g * C + c] = 1.0f / sqrtf(fv + eps);
    if (g == 0) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      if (rmean != nullptr) {
        rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)m;
        rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
      }
      if (rmean2 != nullptr) {
        rmean2[c] = (1.0f - momentum) * rmean2[c] + momentum * (float)m;
        rvar2[c] = (1.0f - momentum) * rvar2[c] + momentum * (float)unb;
      }
    }
  }
}

__global__ void k_norm_finalize(const double* __restrict__ sums, int G, int C, double count, float eps, int mode,
                                float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ rmean,
                                float* __restrict__ rvar, float momentum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * C) return;
  int g = i / C, c = i - g * C;
  double S = sums[(int64_t)g * 2 * C + c], SS = sums[(int64_t)g * 2 * C + C + c];
  double m = S / count;
  double var = (SS - S * m) / count;
  if (var < 0.0) var = 0.0;
  float fv = (float)var;
  float is = (mode == 1) ? 1.0f / sqrtf(fmaxf(fv, eps)) : 1.0f / sqrtf(fv + eps);
  mean[i] = (float)m;
  invstd[i] = is;
  if (rmean != nullptr && g == 0) {
    double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.0f - momentum) * rmean[c] + momentum * (float)m;
    rvar[c] = (1.0f - momentum) * rvar[c] + momentum * (float)unb;
  }
}

// --------------------------------------------------------------------------------- apply fwd
// (gb2, slope2, y2), nullable: a second modulation of the same normalised x, written in the same pass (x read once)
__global__ __launch_bounds__(256) void k_norm_apply_fwd(const float* __restrict__ x, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd,
                                                         const float* __restrict__ gb, float slope, int64_t P, int C,
                                                         int64_t n4, float* __restrict__ y,
                                                         const float* __restrict__ gb2, float slope2,
                                                         float* __restrict__ y2) {
  const int Q = C >> 2;
#pragma unroll 2
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = e / Q;
    const int q = (int)(e - pix * Q);
    const int64_t g = pix / P;
    float4 v = ld4(x + e * 4);
    const float4 m = ld4(mean + g * C + q * 4), r = ld4(invstd + g * C + q * 4);
    v.x = (v.x - m.x) * r.x; v.y = (v.y - m.y) * r.y; v.z = (v.z - m.z) * r.z; v.w = (v.w - m.w) * r.w;
    if (y2 != nullptr) {
      const float4 ga = ld4(gb2 + pix * 2 * C + q * 4), be = ld4(gb2 + pix * 2 * C + C + q * 4);
      float4 u = make_float4(v.x * (1.f + ga.x) + be.x, v.y * (1.f + ga.y) + be.y, v.z * (1.f + ga.z) + be.z,
                             v.w * (1.f + ga.w) + be.w);
      if (slope2 != 1.0f) { u.x = lrelu(u.x, slope2); u.y = lrelu(u.y, slope2); u.z = lrelu(u.z, slope2); u.w = lrelu(u.w, slope2); }
      st4(y2 + e * 4, u);
    }
    if (gb != nullptr) {
      const float4 ga = ld4(gb + pix * 2 * C + q * 4), be = ld4(gb + pix * 2 * C + C + q * 4);
      v.x = v.x * (1.f + ga.x) + be.x; v.y = v.y * (1.f + ga.y) + be.y;
      v.z = v.z * (1.f + ga.z) + be.z; v.w = v.w * (1.f + ga.w) + be.w;
    }
    if (slope != 1.0f) { v.x = lrelu(v.x, slope); v.y = lrelu(v.y, slope); v.z = lrelu(v.z, slope); v.w = lrelu(v.w, slope); }
    st4(y + e * 4, v);
  }
}

// --------------------------------------------------------------------------------- apply bwd
// pass 1: dgb = [dpre * xhat | dpre], partial sums of dn and dn*xhat (same tiling as the stats kernel)
__global__ __launch_bounds__(256) void k_norm_bwd_reduce(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gb,
                                                          const float* __restrict__ yact, float slope, int64_t P, int C,
                                                          int nchunk, float* __restrict__ dgb,
                                                          double* __restrict__ partial, int gcs) {
  // gcs: floats per pixel of gb — 2C for a [gamma | beta] map, C for a gamma-only map (fused forward; needs yact)
  // yact (nullable): the activated output y of the forward.  With it the LeakyReLU gate is read off y's sign (y > 0 iff
  // the pre-activation was) and the beta half of gb is never touched — the fused forward (csg_wino4_conv_part) does not
  // write beta.
  extern __shared__ __attribute__((aligned(16))) double smd[];
  const int tid = threadIdx.x;
  const int chunk = blockIdx.x, g = blockIdx.y;
  const int64_t per = (P + nchunk - 1) / nchunk;
  const int64_t p0 = chunk * per, p1 = min(P, p0 + per);
  const int Q = C >> 2;
  double* out = partial + ((int64_t)g * nchunk + chunk) * 2 * C;
  for (int qb = 0; qb < Q; qb += 256) {
    const int Qc = min(256, Q - qb);
    const int R = 256 / Qc;
    const int q = tid % Qc, rr = tid / Qc;
    d4 s = d4zero(), ss = d4zero();
    if (rr < R) {
      const int co = (qb + q) * 4;
      const float4 m = ld4(mean + (int64_t)g * C + co), r = ld4(invstd + (int64_t)g * C + co);
#pragma unroll 2
      for (int64_t p = p0 + rr; p < p1; p += R) {
        const int64_t pix = (int64_t)g * P + p;
        const float4 xv = ld4(x + pix * C + co);
        float4 d = ld4(dy + pix * C + co);
        float4 xh = make_float4((xv.x - m.x) * r.x, (xv.y - m.y) * r.y, (xv.z - m.z) * r.z, (xv.w - m.w) * r.w);
        float4 dn = d;
        if (gb != nullptr) {
          const float4 ga = ld4(gb + pix * gcs + co);
          if (slope != 1.0f) {
            if (yact != nullptr) {
              const float4 yv = ld4(yact + pix * C + co);
              d.x *= lrelu_g(yv.x, slope); d.y *= lrelu_g(yv.y, slope); d.z *= lrelu_g(yv.z, slope); d.w *= lrelu_g(yv.w, slope);
            } else {
              const float4 be = ld4(gb + pix * gcs + C + co);
              d.x *= lrelu_g(xh.x * (1.f + ga.x) + be.x, slope); d.y *= lrelu_g(xh.y * (1.f + ga.y) + be.y, slope);
              d.z *= lrelu_g(xh.z * (1.f + ga.z) + be.z, slope); d.w *= lrelu_g(xh.w * (1.f + ga.w) + be.w, slope);
            }
          }
          st4(dgb + pix * 2 * C + co, make_float4(d.x * xh.x, d.y * xh.y, d.z * xh.z, d.w * xh.w));
          st4(dgb + pix * 2 * C + C + co, d);
          dn = make_float4(d.x * (1.f + ga.x), d.y * (1.f + ga.y), d.z * (1.f + ga.z), d.w * (1.f + ga.w));
        } else if (slope != 1.0f) {
          dn.x *= lrelu_g(xh.x, slope); dn.y *= lrelu_g(xh.y, slope); dn.z *= lrelu_g(xh.z, slope); dn.w *= lrelu_g(xh.w, slope);
        }
        s.x += dn.x; s.y += dn.y; s.z += dn.z; s.w += dn.w;
        ss.x += (double)dn.x * xh.x; ss.y += (double)dn.y * xh.y; ss.z += (double)dn.z * xh.z; ss.w += (double)dn.w * xh.w;
      }
      double* o = &smd[(rr * Qc + q) * 8];
      o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
      o[4] = ss.x; o[5] = ss.y; o[6] = ss.z; o[7] = ss.w;
    }
    __syncthreads();
    if (tid < Qc) {
      double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += smd[(r * Qc + tid) * 8 + e];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        out[(qb + tid) * 4 + e] = a[e];
        out[C + (qb + tid) * 4 + e] = a[4 + e];
      }
    }
    __syncthreads();
  }
}

// pass 2: dx = invstd * (dn - mean(dn) - xhat * mean(dn*xhat))
// With (dy2, gb2) a SECOND SPADE modulation of the same normalised x (a residual block's norm_0 and norm_s: same batch
// statistics) is folded in: dn = dn_1 + dn_2, `dsums` already holds the sum of both reductions — one pass and one dx
// instead of two passes and an addition.
__global__ __launch_bounds__(256) void k_norm_bwd_dx(const float* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gb, float slope,
                                                      const double* __restrict__ dsums, double inv_count, int64_t P,
                                                      int C, int64_t n4, float* __restrict__ dx,
                                                      const float* __restrict__ dy2, const float* __restrict__ gb2,
                                                      float slope2, const float* __restrict__ dgb,
                                                      const float* __restrict__ dgb2, int gcs) {
  const int Q = C >> 2;
#pragma unroll 2
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = e / Q;
    const int q = (int)(e - pix * Q);
    const int64_t g = pix / P;
    const int co = q * 4;
    const float4 xv = ld4(x + e * 4);
    float4 d = (gb != nullptr && dgb != nullptr) ? make_float4(0.f, 0.f, 0.f, 0.f) : ld4(dy + e * 4);
    const float4 m = ld4(mean + g * C + co), r = ld4(invstd + g * C + co);
    const float4 xh = make_float4((xv.x - m.x) * r.x, (xv.y - m.y) * r.y, (xv.z - m.z) * r.z, (xv.w - m.w) * r.w);
    float4 dn = d;
    if (gb != nullptr && dgb != nullptr) {
      // pass 1 already wrote d(beta) = dy * activation gate: read it instead of dy and beta (one map less)
      const float4 ga = ld4(gb + pix * gcs + co);
      d = ld4(dgb + pix * 2 * C + C + co);
      dn = make_float4(d.x * (1.f + ga.x), d.y * (1.f + ga.y), d.z * (1.f + ga.z), d.w * (1.f + ga.w));
    } else if (gb != nullptr) {
      const float4 ga = ld4(gb + pix * gcs + co);
      if (slope != 1.0f) {
        const float4 be = ld4(gb + pix * gcs + C + co);
        d.x *= lrelu_g(xh.x * (1.f + ga.x) + be.x, slope); d.y *= lrelu_g(xh.y * (1.f + ga.y) + be.y, slope);
        d.z *= lrelu_g(xh.z * (1.f + ga.z) + be.z, slope); d.w *= lrelu_g(xh.w * (1.f + ga.w) + be.w, slope);
      }
      dn = make_float4(d.x * (1.f + ga.x), d.y * (1.f + ga.y), d.z * (1.f + ga.z), d.w * (1.f + ga.w));
    } else if (slope != 1.0f) {
      dn.x *= lrelu_g(xh.x, slope); dn.y *= lrelu_g(xh.y, slope); dn.z *= lrelu_g(xh.z, slope); dn.w *= lrelu_g(xh.w, slope);
    }
    if (dy2 != nullptr) {                 // second modulation (gb2 is required with it)
      const float4 ga = ld4(gb2 + pix * gcs + co);
      float4 d2 = dgb2 != nullptr ? ld4(dgb2 + pix * 2 * C + C + co) : ld4(dy2 + e * 4);
      if (slope2 != 1.0f && dgb2 == nullptr) {
        const float4 be = ld4(gb2 + pix * gcs + C + co);
        d2.x *= lrelu_g(xh.x * (1.f + ga.x) + be.x, slope2); d2.y *= lrelu_g(xh.y * (1.f + ga.y) + be.y, slope2);
        d2.z *= lrelu_g(xh.z * (1.f + ga.z) + be.z, slope2); d2.w *= lrelu_g(xh.w * (1.f + ga.w) + be.w, slope2);
      }
      dn.x += d2.x * (1.f + ga.x); dn.y += d2.y * (1.f + ga.y); dn.z += d2.z * (1.f + ga.z); dn.w += d2.w * (1.f + ga.w);
    }
    const double* ds = dsums + g * 2 * C;
    float4 a = make_float4((float)(ds[co] * inv_count), (float)(ds[co + 1] * inv_count), (float)(ds[co + 2] * inv_count),
                           (float)(ds[co + 3] * inv_count));
    float4 b = make_float4((float)(ds[C + co] * inv_count), (float)(ds[C + co + 1] * inv_count),
                           (float)(ds[C + co + 2] * inv_count), (float)(ds[C + co + 3] * inv_count));
    float4 o;
    o.x = r.x * (dn.x - a.x - xh.x * b.x); o.y = r.y * (dn.y - a.y - xh.y * b.y);
    o.z = r.z * (dn.z - a.z - xh.z * b.z); o.w = r.w * (dn.w - a.w - xh.w * b.w);
    st4(dx + e * 4, o);
  }
}

// --------------------------------------------------------------------------------- elementwise
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, int64_t n, int act, float slope,
                          float* __restrict__ dpre) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float g = dy[i], v = y[i];
    if (act == CSG_ACT_LEAKY)
      g *= (v > 0.f ? 1.f : slope);
    else if (act == CSG_ACT_TANH)
      g *= (1.f - v * v);
    dpre[i] = g;
  }
}

// F.interpolate(x, size=(OH, OW), mode='nearest') on NHWC maps (reference normalization.py:98: the segmentation map is
// resized to every SPADE layer's resolution): source index = min(floor(dst * (in / out)), in - 1) with the scale in
// fp32, as ATen computes it.
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) { return min((int)floorf(dst * scale), in - 1); }

__global__ void k_nearest_fwd(const float* __restrict__ x, int IH, int IW, int OH, int OW, int Q, float sy, float sx,
                              int64_t n4, float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = e / Q;
    const int q = (int)(e - pix * Q);
    const int X = (int)(pix % OW);
    const int64_t t = pix / OW;
    const int Y = (int)(t % OH);
    const int64_t b = t / OH;
    st4(y + e * 4, ld4(x + (((b * IH + nearest_src(Y, sy, IH)) * IW + nearest_src(X, sx, IW)) * (int64_t)Q + q) * 4));
  }
}

// dx[iy][ix] = sum of dy over the output pixels that read (iy, ix): a thread per input element walks its (contiguous)
// range of output rows and columns in order — no atomics, bit-reproducible
__global__ void k_nearest_bwd(const float* __restrict__ dy, int IH, int IW, int OH, int OW, int Q, float sy, float sx,
                              int64_t n4, float* __restrict__ dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = e / Q;
    const int q = (int)(e - pix * Q);
    const int ix = (int)(pix % IW);
    const int64_t t = pix / IW;
    const int iy = (int)(t % IH);
    const int64_t b = t / IH;
    int y0 = max(0, (int)((float)iy / sy) - 2), x0 = max(0, (int)((float)ix / sx) - 2);
    while (y0 < OH && nearest_src(y0, sy, IH) < iy) ++y0;
    while (x0 < OW && nearest_src(x0, sx, IW) < ix) ++x0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int Y = y0; Y < OH && nearest_src(Y, sy, IH) == iy; ++Y)
      for (int X = x0; X < OW && nearest_src(X, sx, IW) == ix; ++X) {
        const float4 g = ld4(dy + (((b * OH + Y) * OW + X) * (int64_t)Q + q) * 4);
        acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
      }
    st4(dx + e * 4, acc);
  }
}

__global__ void k_upsample2x_fwd(const float* __restrict__ x, int H, int W, int Q, int64_t n4, float* __restrict__ y) {
  // y is (B, 2H, 2W, C)
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int X = (int)(pix % (2 * W));
    int64_t t = pix / (2 * W);
    int Y = (int)(t % (2 * H));
    int64_t b = t / (2 * H);
    st4(y + e * 4, ld4(x + (((b * H + (Y >> 1)) * W + (X >> 1)) * (int64_t)Q + q) * 4));
  }
}

__global__ void k_upsample2x_bwd(const float* __restrict__ dy, int H, int W, int Q, int64_t n4,
                                 float* __restrict__ dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int xw = (int)(pix % W);
    int64_t t = pix / W;
    int yh = (int)(t % H);
    int64_t b = t / H;
    const float* base = dy + (((b * 2 * H + 2 * yh) * 2 * W + 2 * xw) * (int64_t)Q + q) * 4;
    const int64_t rs = (int64_t)2 * W * Q * 4;
    float4 a = ld4(base), c = ld4(base + Q * 4), d = ld4(base + rs), f = ld4(base + rs + Q * 4);
    st4(dx + e * 4, make_float4(a.x + c.x + d.x + f.x, a.y + c.y + d.y + f.y, a.z + c.z + d.z + f.z, a.w + c.w + d.w + f.w));
  }
}

__global__ void k_avgpool3s2_fwd(const float* __restrict__ x, int H, int W, int OH, int OW, int Q, int64_t n4,
                                 float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int ox = (int)(pix % OW);
    int64_t t = pix / OW;
    int oy = (int)(t % OH);
    int64_t b = t / OH;
    float4 a = f4(0.f);
    int cnt = 0;
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float4 v = ld4(x + (((b * H + iy) * W + ix) * (int64_t)Q + q) * 4);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        ++cnt;
      }
    }
    float inv = 1.0f / (float)cnt;
    st4(y + e * 4, make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv));
  }
}

__device__ __forceinline__ int win_count(int o, int n) {  // in-bounds taps of a 3-wide window at stride 2, pad 1
  int lo = o * 2 - 1, hi = o * 2 + 1;
  if (lo < 0) lo = 0;
  if (hi > n - 1) hi = n - 1;
  return hi - lo + 1;
}

// `add` (nullable, may alias dx): dx = add + pool_bwd(dy) — the input fans out to a consumer of its own AND to the pooled
// copy (the PatchGAN's two scales), and its gradient is the sum of both; the pooled part is summed first, as autograd's
// separate addition of the two tensors did (a two-term fp32 sum does not depend on the order of its terms).
__global__ void k_avgpool3s2_bwd(const float* __restrict__ dy, int H, int W, int OH, int OW, int Q, int64_t n4,
                                 const float* add, float* dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int ix = (int)(pix % W);
    int64_t t = pix / W;
    int iy = (int)(t % H);
    int64_t b = t / H;
    float4 a = f4(0.f);
    // outputs oy with |2*oy - iy| <= 1
    for (int oy = iy / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= OH) continue;
      int cy = win_count(oy, H);
      for (int ox = ix / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= OW) continue;
        float inv = 1.0f / (float)(cy * win_count(ox, W));
        float4 v = ld4(dy + (((b * OH + oy) * OW + ox) * (int64_t)Q + q) * 4);
        a.x += v.x * inv; a.y += v.y * inv; a.z += v.z * inv; a.w += v.w * inv;
      }
    }
    if (add != nullptr) {
      const float4 o = ld4(add + e * 4);
      a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
    }
    st4(dx + e * 4, a);
  }
}

// --------------------------------------------------------------------------------- C ABI
static inline unsigned ew_grid(int64_t n) {
  int64_t g = cdiv(n, 256);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

extern "C" {

int csg_norm_stats(const float* x, int64_t G, int64_t P, int64_t C, double* sums, double* partial, int64_t nchunk,
                   void* stream) {
  CSG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_norm_stats: bad shape G=%ld P=%ld C=%ld",
              (long)G, (long)P, (long)C);
  CSG_REQUIRE(nchunk >= 1 && nchunk <= 65535 && G <= 65535, CSG_E_BADSHAPE, "csg_norm_stats: bad nchunk");
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p(K_NORM_STATS, (double)G * P * C * 4, s);
    CSG_LAUNCH(k_norm_stats_partial, dim3((unsigned)nchunk, (unsigned)G), dim3(256), 256 * 8 * 8, s, x, P,
                       (int)C, C, (int)nchunk, partial);
  }
  CSG_LAUNCH(k_partial_reduce<double>, dim3((unsigned)cdiv(2 * C, 32), (unsigned)G), dim3(256), 0, s, partial,
                     (int)(2 * C), (int)nchunk, (int)(2 * C), sums);
  return check_launch("csg_norm_stats");
}

int csg_norm_stats_finalize(const float* x, int64_t G, int64_t P, int64_t C, double* partial, int64_t nchunk, double count,
                            float eps, float* mean, float* invstd, float* running_mean, float* running_var,
                            float* running_mean2, float* running_var2, float momentum, void* stream) {
  CSG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0 && count > 0, CSG_E_BADSHAPE,
              "csg_norm_stats_finalize: bad shape G=%ld P=%ld C=%ld", (long)G, (long)P, (long)C);
  CSG_REQUIRE(nchunk >= 1 && nchunk <= 65535 && G <= 65535, CSG_E_BADSHAPE, "csg_norm_stats_finalize: bad nchunk");
  CSG_REQUIRE((running_mean == nullptr && running_mean2 == nullptr) || G == 1, CSG_E_UNSUPPORTED,
              "csg_norm_stats_finalize: running stats need G == 1");
  CSG_REQUIRE((running_mean == nullptr) == (running_var == nullptr) && (running_mean2 == nullptr) == (running_var2 == nullptr),
              CSG_E_BADSHAPE, "csg_norm_stats_finalize: running mean and variance come together");
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p(K_NORM_STATS, (double)G * P * C * 4, s);
    CSG_LAUNCH(k_norm_stats_partial, dim3((unsigned)nchunk, (unsigned)G), dim3(256), 256 * 8 * 8, s, x, P, (int)C, C,
               (int)nchunk, partial);
  }
  ProfScope p(K_NORM_FINALIZE, (double)G * C * 24, s);
  CSG_LAUNCH(k_partial_reduce_finalize, dim3((unsigned)cdiv(C, 16), (unsigned)G), dim3(256), 0, s, partial, (int)C, (int)nchunk,
             count, eps, mean, invstd, running_mean, running_var, running_mean2, running_var2, momentum);
  return check_launch("csg_norm_stats_finalize");
}

int csg_norm_finalize(const double* sums, int64_t G, int64_t C, double count, float eps, int32_t mode, float* mean,
                      float* invstd, float* running_mean, float* running_var, float momentum, void* stream) {
  CSG_REQUIRE(G > 0 && C > 0 && count > 0, CSG_E_BADSHAPE, "csg_norm_finalize: bad shape");
  CSG_REQUIRE(running_mean == nullptr || G == 1, CSG_E_UNSUPPORTED, "csg_norm_finalize: running stats need G == 1");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_NORM_FINALIZE, (double)G * C * 24, s);
  CSG_LAUNCH(k_norm_finalize, dim3((unsigned)cdiv(G * C, 256)), dim3(256), 0, s, sums, (int)G, (int)C, count,
                     eps, mode, mean, invstd, running_mean, running_var, momentum);
  return check_launch("csg_norm_finalize");
}

int csg_norm_apply_fwd(const float* x, const float* mean, const float* invstd, const float* gb, float slope,
                       int64_t G, int64_t P, int64_t C, float* y, const float* gb2, float slope2, float* y2,
                       void* stream) {
  CSG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_norm_apply_fwd: bad shape");
  CSG_REQUIRE((gb2 == nullptr) == (y2 == nullptr), CSG_E_BADSHAPE, "csg_norm_apply_fwd: gb2 and y2 come together");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = G * P * C / 4;
  ProfScope p(K_NORM_APPLY_FWD, (double)G * P * C * 4 * ((gb ? 4 : 2) + (y2 ? 3 : 0)), s);
  CSG_LAUNCH(k_norm_apply_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, mean, invstd, gb, slope, P, (int)C, n4, y, gb2,
                     slope2, y2);
  return check_launch("csg_norm_apply_fwd");
}

int csg_norm_apply_bwd_reduce(const float* dy, const float* x, const float* mean, const float* invstd,
                              const float* gb, const float* yact, float slope, int64_t G, int64_t P, int64_t C, float* dgb,
                              double* dsums, double* partial, int64_t nchunk, int64_t gb_cs, void* stream) {
  CSG_REQUIRE(yact == nullptr || gb != nullptr, CSG_E_BADSHAPE, "csg_norm_apply_bwd_reduce: yact goes with a modulation");
  CSG_REQUIRE(gb == nullptr || gb_cs == 2 * C || (gb_cs == C && yact != nullptr), CSG_E_BADSHAPE,
              "csg_norm_apply_bwd_reduce: gb_cs is 2C ([gamma | beta]) or, with yact, C (gamma only)");
  CSG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_norm_apply_bwd_reduce: bad shape");
  CSG_REQUIRE((gb == nullptr) == (dgb == nullptr), CSG_E_BADSHAPE, "csg_norm_apply_bwd_reduce: gb/dgb mismatch");
  CSG_REQUIRE(nchunk >= 1 && nchunk <= 65535 && G <= 65535, CSG_E_BADSHAPE, "csg_norm_apply_bwd_reduce: bad nchunk");
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope p(K_NORM_BWD_REDUCE, (double)G * P * C * 4 * (gb ? 6 : 2), s);
    CSG_LAUNCH(k_norm_bwd_reduce, dim3((unsigned)nchunk, (unsigned)G), dim3(256), 256 * 8 * 8, s, dy, x, mean,
                       invstd, gb, yact, slope, P, (int)C, (int)nchunk, dgb, partial, (int)gb_cs);
  }
  CSG_LAUNCH(k_partial_reduce<double>, dim3((unsigned)cdiv(2 * C, 32), (unsigned)G), dim3(256), 0, s, partial,
                     (int)(2 * C), (int)nchunk, (int)(2 * C), dsums);
  return check_launch("csg_norm_apply_bwd_reduce");
}

int csg_norm_apply_bwd_dx(const float* dy, const float* x, const float* mean, const float* invstd, const float* gb,
                          float slope, const double* dsums, double count, int64_t G, int64_t P, int64_t C, float* dx,
                          const float* dy2, const float* gb2, float slope2, const float* dgb, const float* dgb2,
                          int64_t gb_cs, void* stream) {
  CSG_REQUIRE(G > 0 && P > 0 && C > 0 && C % 4 == 0 && count > 0, CSG_E_BADSHAPE, "csg_norm_apply_bwd_dx: bad shape");
  CSG_REQUIRE(gb == nullptr || gb_cs == 2 * C || (gb_cs == C && dgb != nullptr && (gb2 == nullptr || dgb2 != nullptr)),
              CSG_E_BADSHAPE, "csg_norm_apply_bwd_dx: gb_cs is 2C ([gamma | beta]) or, with dgb (and dgb2), C (gamma only)");
  CSG_REQUIRE((dy2 == nullptr) == (gb2 == nullptr), CSG_E_BADSHAPE, "csg_norm_apply_bwd_dx: dy2 and gb2 come together");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = G * P * C / 4;
  ProfScope p(K_NORM_BWD_DX, (double)G * P * C * 4 * ((gb ? (dgb ? 4 : 5) : 3) + (dy2 ? (dgb2 ? 2 : 3) : 0)), s);
  CSG_LAUNCH(k_norm_bwd_dx, dim3(ew_grid(n4)), dim3(256), 0, s, dy, x, mean, invstd, gb, slope, dsums,
                     1.0 / count, P, (int)C, n4, dx, dy2, gb2, slope2, dgb, dgb2, (int)gb_cs);
  return check_launch("csg_norm_apply_bwd_dx");
}

int csg_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float slope, float* dpre, void* stream) {
  CSG_REQUIRE(n >= 0, CSG_E_BADSHAPE, "csg_act_bwd: bad n");
  if (n == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_ACT_BWD, (double)n * 12, s);
  CSG_LAUNCH(k_act_bwd, dim3(ew_grid(n)), dim3(256), 0, s, dy, y, n, act, slope, dpre);
  return check_launch("csg_act_bwd");
}

int csg_colsum(const float* x, int64_t rows, int64_t C, int64_t x_cs, float* out, double* partial, int64_t nchunk,
               void* stream) {
  CSG_REQUIRE(rows > 0 && C > 0 && C % 4 == 0 && x_cs % 4 == 0, CSG_E_BADSHAPE, "csg_colsum: bad shape");
  CSG_REQUIRE(nchunk >= 1 && nchunk <= 65535, CSG_E_BADSHAPE, "csg_colsum: bad nchunk");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_COLSUM, (double)rows * C * 4, s);
  CSG_LAUNCH(k_norm_stats_partial, dim3((unsigned)nchunk, 1), dim3(256), 256 * 8 * 8, s, x, rows, (int)C, x_cs,
                     (int)nchunk, partial);
  CSG_LAUNCH(k_partial_reduce<float>, dim3((unsigned)cdiv(C, 32), 1), dim3(256), 0, s, partial, (int)(2 * C),
                     (int)nchunk, (int)C, out);
  return check_launch("csg_colsum");
}

int csg_nearest_resize_fwd(const float* x, int64_t B, int64_t IH, int64_t IW, int64_t C, int64_t OH, int64_t OW, float* y,
                           void* stream) {
  CSG_REQUIRE(B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_nearest_resize_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = B * OH * OW * C / 4;
  ProfScope p(K_UPSAMPLE_FWD, (double)n4 * 32, s);
  CSG_LAUNCH(k_nearest_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, (int)IH, (int)IW, (int)OH, (int)OW, (int)(C / 4),
             (float)IH / (float)OH, (float)IW / (float)OW, n4, y);
  return check_launch("csg_nearest_resize_fwd");
}

int csg_nearest_resize_bwd(const float* dy, int64_t B, int64_t IH, int64_t IW, int64_t C, int64_t OH, int64_t OW, float* dx,
                           void* stream) {
  CSG_REQUIRE(B > 0 && IH > 0 && IW > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_nearest_resize_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = B * IH * IW * C / 4;
  ProfScope p(K_UPSAMPLE_BWD, (double)(n4 + B * OH * OW * C / 4) * 16, s);
  CSG_LAUNCH(k_nearest_bwd, dim3(ew_grid(n4)), dim3(256), 0, s, dy, (int)IH, (int)IW, (int)OH, (int)OW, (int)(C / 4),
             (float)IH / (float)OH, (float)IW / (float)OW, n4, dx);
  return check_launch("csg_nearest_resize_bwd");
}

int csg_upsample2x_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_upsample2x_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = B * 4 * H * W * C / 4;
  ProfScope p(K_UPSAMPLE_FWD, (double)n4 * 16 * 1.25, s);
  CSG_LAUNCH(k_upsample2x_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, (int)H, (int)W, (int)(C / 4), n4, y);
  return check_launch("csg_upsample2x_fwd");
}

int csg_upsample2x_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_upsample2x_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t n4 = B * H * W * C / 4;
  ProfScope p(K_UPSAMPLE_BWD, (double)n4 * 16 * 5, s);
  CSG_LAUNCH(k_upsample2x_bwd, dim3(ew_grid(n4)), dim3(256), 0, s, dy, (int)H, (int)W, (int)(C / 4), n4, dx);
  return check_launch("csg_upsample2x_bwd");
}

int csg_avgpool3s2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "csg_avgpool3s2_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t n4 = B * OH * OW * C / 4;
  ProfScope p(K_AVGPOOL_FWD, (double)(B * H * W * C + n4 * 4) * 4, s);
  CSG_LAUNCH(k_avgpool3s2_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, (int)H, (int)W, (int)OH, (int)OW,
                     (int)(C / 4), n4, y);
  return check_launch("csg_avgpool3s2_fwd");
}

static int avgpool3s2_bwd_launch(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, const float* add, float* dx,
                                 void* stream, const char* who) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE, "%s: bad shape", who);
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const int64_t n4 = B * H * W * C / 4;
  ProfScope p(K_AVGPOOL_BWD, (double)(B * H * W * C * (add != nullptr ? 2 : 1) + B * OH * OW * C) * 4, s);
  CSG_LAUNCH(k_avgpool3s2_bwd, dim3(ew_grid(n4)), dim3(256), 0, s, dy, (int)H, (int)W, (int)OH, (int)OW,
                     (int)(C / 4), n4, add, dx);
  return check_launch(who);
}

int csg_avgpool3s2_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream) {
  return avgpool3s2_bwd_launch(dy, B, H, W, C, nullptr, dx, stream, "csg_avgpool3s2_bwd");
}

int csg_avgpool3s2_bwd_add(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, const float* add, float* dx,
                           void* stream) {
  CSG_REQUIRE(add != nullptr && ((uintptr_t)add % 16) == 0, CSG_E_BADSHAPE, "csg_avgpool3s2_bwd_add: needs a 16-byte aligned addend");
  return avgpool3s2_bwd_launch(dy, B, H, W, C, add, dx, stream, "csg_avgpool3s2_bwd_add");
}

}  // extern "C"
