// Spectral normalisation of a conv weight — the forward pre-hook `torch.nn.utils.spectral_norm` runs
// before EVERY call of the 24 generator and 6 discriminator convolutions that carry it
// (reference call sites: spade/models/networks/architecture.py:35-39, normalization.py:27; algorithm:
// torch/nn/utils/spectral_norm.py `SpectralNorm.compute_weight`, n_power_iterations = 1, dim = 0):
//
//     v <- normalize(W^T u);  u <- normalize(W v)      (training mode only, buffers updated in place)
//     sigma = u . (W v);      W_eff = W / sigma
//
// with W = weight_orig.view(Cout, -1) and normalize(x) = x / max(||x||, eps).  PyTorch issues ~15 tiny
// kernels per call (two GEMVs, norms, clamps, divisions, clones, a dot); here it is four streaming
// launches over W, and two for the backward  dW = (dW_eff - (sum dW_eff.W_eff) u v^T) / sigma
// (u, v are constants of the graph, as in PyTorch where they are computed under no_grad).
// HBM bound: W is read three times forward (37.7 MB for the 1024x1024x3x3 layers) and twice backward.
// All reductions are two-stage with a fixed combination order (bit-reproducible).
#include "csg_common.h"

using namespace csg;

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }

// block-wide sum of doubles, 1024 threads max; result valid in every thread
__device__ double block_sum(double v, double* sm) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sm[wv] = v;
  __syncthreads();
  double t = 0.0;
  for (int i = 0; i < nw; ++i) t += sm[i];
  return t;
}

// part[r][k] = sum over the rows of chunk r of u[i] * W[i][k]          grid (ceil(K/1024), R)
__device__ __forceinline__ void d_sn_wtu_partial(const float* __restrict__ w, const float* __restrict__ u, int Cout, int K,
                                                 int R, float* __restrict__ part, int bx, int by) {
  const int k = (bx * 256 + threadIdx.x) * 4;
  if (k >= K) return;
  const int r = by;
  const int per = (Cout + R - 1) / R;
  const int i0 = r * per, i1 = min(Cout, i0 + per);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  int i = i0;
  for (; i + 3 < i1; i += 4) {
    const float u0 = u[i], u1 = u[i + 1], u2 = u[i + 2], u3 = u[i + 3];
    const float4 w0 = ld4(w + (int64_t)i * K + k), w1 = ld4(w + (int64_t)(i + 1) * K + k),
                 w2 = ld4(w + (int64_t)(i + 2) * K + k), w3 = ld4(w + (int64_t)(i + 3) * K + k);
    a.x += u0 * w0.x; a.y += u0 * w0.y; a.z += u0 * w0.z; a.w += u0 * w0.w;
    a.x += u1 * w1.x; a.y += u1 * w1.y; a.z += u1 * w1.z; a.w += u1 * w1.w;
    a.x += u2 * w2.x; a.y += u2 * w2.y; a.z += u2 * w2.z; a.w += u2 * w2.w;
    a.x += u3 * w3.x; a.y += u3 * w3.y; a.z += u3 * w3.z; a.w += u3 * w3.w;
  }
  for (; i < i1; ++i) {
    const float ui = u[i];
    const float4 wi = ld4(w + (int64_t)i * K + k);
    a.x += ui * wi.x; a.y += ui * wi.y; a.z += ui * wi.z; a.w += ui * wi.w;
  }
  *(float4*)(part + (int64_t)r * K + k) = a;
}

__global__ __launch_bounds__(256) void k_sn_wtu_partial(const float* __restrict__ w, const float* __restrict__ u,
                                                         int Cout, int K, int R, float* __restrict__ part) {
  d_sn_wtu_partial(w, u, Cout, K, R, part, blockIdx.x, blockIdx.y);
}

// t[k] = sum_r part[r][k]
__device__ __forceinline__ void d_sn_wtu_reduce(const float* __restrict__ part, int K, int R, float* __restrict__ t, int bx) {
  const int k = (bx * blockDim.x + threadIdx.x) * 4;
  if (k >= K) return;
  float4 a = ld4(part + k);
#pragma unroll 8
  for (int r = 1; r < R; ++r) {
    const float4 b = ld4(part + (int64_t)r * K + k);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
  }
  *(float4*)(t + k) = a;
}

__global__ void k_sn_wtu_reduce(const float* __restrict__ part, int K, int R, float* __restrict__ t) {
  d_sn_wtu_reduce(part, K, R, t, blockIdx.x);
}

// s[i] = W[i] . v with v = t / max(||t||, eps) formed on the fly (training) or v = the stored buffer (eval).
// One block per row (all of a thread's loads are independent: one HBM round trip per row); every block
// recomputes ||t|| in the same order (t is K floats, L2 resident), which saves a single-block
// normalisation launch.  Block 0 publishes v.
__device__ __forceinline__ void d_sn_rowdot(const float* __restrict__ w, const float* __restrict__ t, int Cout, int K,
                                            float eps, int iterate, float* __restrict__ v, float* __restrict__ v_used,
                                            float* __restrict__ s, int row, double* sm) {
  const int tid = threadIdx.x;
  const float* vin = iterate ? t : v;
  float nrm = 1.f;
  if (iterate) {
    double acc = 0.0;
    for (int k = tid * 4; k < K; k += 1024) {
      const float4 a = ld4(t + k);
      acc += ((double)a.x * a.x + (double)a.y * a.y) + ((double)a.z * a.z + (double)a.w * a.w);
    }
    nrm = fmaxf((float)sqrt(block_sum(acc, sm)), eps);
  }
  const bool publish = row == 0;
  const float* wr = w + (int64_t)row * K;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 4
  for (int k = tid * 4; k < K; k += 1024) {
    const float4 a = ld4(wr + k);
    float4 b = ld4(vin + k);
    if (iterate) b = make_float4(b.x / nrm, b.y / nrm, b.z / nrm, b.w / nrm);
    if (publish) {
      if (iterate) *(float4*)(v + k) = b;
      *(float4*)(v_used + k) = b;
    }
    acc0 += a.x * b.x + a.y * b.y;
    acc1 += a.z * b.z + a.w * b.w;
  }
  const double tot = block_sum((double)(acc0 + acc1), sm);
  if (tid == 0) s[row] = (float)tot;
}

__global__ __launch_bounds__(256) void k_sn_rowdot(const float* __restrict__ w, const float* __restrict__ t,
                                                    int Cout, int K, float eps, int iterate, float* __restrict__ v,
                                                    float* __restrict__ v_used, float* __restrict__ s) {
  __shared__ double sm[16];
  d_sn_rowdot(w, t, Cout, K, eps, iterate, v, v_used, s, blockIdx.x, sm);
}

// W_eff = W / sigma.  Every block derives sigma from s (Cout floats) in the same order:
// training: u = s / max(||s||, eps), sigma = u . s;  eval: sigma = u_stored . s.  Block 0 publishes u, sigma.
__device__ __forceinline__ void d_sn_scale(const float* __restrict__ w, const float* __restrict__ s, int Cout, float eps,
                                           int iterate, float* __restrict__ u, float* __restrict__ u_used,
                                           float* __restrict__ sigma, int64_t n4, float* __restrict__ w_eff, int Cin, int khw,
                                           int bx, int gx, double* sm, float* row) {
  float nrm = 1.f;
  if (iterate) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < Cout; i += 256) acc += (double)s[i] * (double)s[i];
    nrm = fmaxf((float)sqrt(block_sum(acc, sm)), eps);
  }
  double dot = 0.0;
  for (int i = threadIdx.x; i < Cout; i += 256) {
    const float ui = iterate ? s[i] / nrm : u[i];
    if (bx == 0) {
      if (iterate) u[i] = ui;
      u_used[i] = ui;
    }
    dot += (double)ui * (double)s[i];
  }
  const float sg = (float)block_sum(dot, sm);
  if (bx == 0 && threadIdx.x == 0) sigma[0] = sg;
  if (Cin == 0) {                        // W_eff in the memory order of W
    for (int64_t e = (int64_t)bx * 256 + threadIdx.x; e < n4; e += (int64_t)gx * 256) {
      const float4 a = ld4(w + e * 4);
      *(float4*)(w_eff + e * 4) = make_float4(a.x / sg, a.y / sg, a.z / sg, a.w / sg);     // `weight / sigma`
    }
    return;
  }
  // W is (Cout, Cin, KH, KW) row-major; W_eff is written in channels-last memory [Cout][KH][KW][Cin] — the
  // convolution kernels' forward operand.  One block per output channel: its row is read as it lies, transposed
  // (cin, tap) -> (tap, cin) through LDS and written as it will lie (both sides coalesced).
  const int K = Cin * khw, LD = Cin + 4;
  for (int co = bx; co < Cout; co += gx) {
    __syncthreads();
    for (int e = threadIdx.x; e < (K >> 2); e += 256) {
      const float4 a = ld4(w + (int64_t)co * K + e * 4);
      const float v[4] = {a.x / sg, a.y / sg, a.z / sg, a.w / sg};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * e + j, ci = k / khw, t = k - ci * khw;
        row[t * LD + ci] = v[j];
      }
    }
    __syncthreads();
    const int q = Cin >> 2;
    for (int e = threadIdx.x; e < (K >> 2); e += 256) {
      const int t = e / q, c4 = e - t * q;
      *(float4*)(w_eff + (int64_t)co * K + e * 4) = *(const float4*)(row + t * LD + c4 * 4);
    }
  }
}

__global__ __launch_bounds__(256) void k_sn_scale(const float* __restrict__ w, const float* __restrict__ s, int Cout,
                                                   float eps, int iterate, float* __restrict__ u,
                                                   float* __restrict__ u_used, float* __restrict__ sigma, int64_t n4,
                                                   float* __restrict__ w_eff, int Cin, int khw) {
  __shared__ double sm[16];
  extern __shared__ __attribute__((aligned(16))) float row[];          // [khw][Cin + 4]
  d_sn_scale(w, s, Cout, eps, iterate, u, u_used, sigma, n4, w_eff, Cin, khw, blockIdx.x, gridDim.x, sm, row);
}

struct SnGeom {
  int Cout, Cin, KH, KW;
  int64_t s0, s1, s2, s3;   // element strides of dW_eff along (Cout, Cin, KH, KW)
};

// offset of element k = (ci, kh, kw) of a row inside the (dense) dW_eff row
__device__ __forceinline__ int row_off(const SnGeom& g, int k) {
  const int khw = g.KH * g.KW;
  const int ci = k / khw, rem = k - ci * khw;
  const int kh = rem / g.KW, kw = rem - kh * g.KW;
  return (int)(ci * g.s1 + kh * g.s2 + kw * g.s3);
}

// One block per output channel.  dW_eff usually arrives in the weight-gradient kernel's layout
// [Cout][KH][KW][Cin] while W is [Cout][Cin][KH][KW]: the row is staged in LDS in its own memory order
// (coalesced) and read back permuted, so both global streams stay coalesced.
// partial[row] = sum_k dW_eff[row][k] * W[row][k]
__device__ __forceinline__ void d_sn_bwd_dot(const SnGeom& g, const float* __restrict__ dweff, const float* __restrict__ w,
                                             int K, double* __restrict__ partial, int co, double* sm, float* row) {
  const float* src = dweff + (int64_t)co * g.s0;
  for (int j = threadIdx.x * 4; j < K; j += 256 * 4) *(float4*)(row + j) = ld4(src + j);
  __syncthreads();
  const float* wr = w + (int64_t)co * K;
  double acc = 0.0;
  for (int k = threadIdx.x; k < K; k += 256) acc += (double)(row[row_off(g, k)] * wr[k]);
  acc = block_sum(acc, sm);
  if (threadIdx.x == 0) partial[co] = acc;
}

__global__ __launch_bounds__(256) void k_sn_bwd_dot(SnGeom g, const float* __restrict__ dweff,
                                                     const float* __restrict__ w, int K,
                                                     double* __restrict__ partial) {
  extern __shared__ float row[];
  __shared__ double sm[16];
  d_sn_bwd_dot(g, dweff, w, K, partial, blockIdx.x, sm, row);
}

// dW = (dW_eff - c u v^T) / sigma,  c = sum dW_eff . W_eff = (sum dW_eff . W) / sigma
__device__ __forceinline__ void d_sn_bwd_dw(const SnGeom& g, const float* __restrict__ dweff, const float* __restrict__ u,
                                            const float* __restrict__ v, const float* __restrict__ sigma,
                                            const double* __restrict__ partial, int K, float* __restrict__ dw, int co,
                                            double* sm, float* row) {
  const float* src = dweff + (int64_t)co * g.s0;
  for (int j = threadIdx.x * 4; j < K; j += 256 * 4) *(float4*)(row + j) = ld4(src + j);
  double tot = 0.0;
  for (int i = threadIdx.x; i < g.Cout; i += 256) tot += partial[i];
  tot = block_sum(tot, sm);                      // same order in every block: one value of c (syncs the LDS row too)
  const float sg = sigma[0];
  const float cu = (float)(tot / (double)sg) * u[co];
  float* dr = dw + (int64_t)co * K;
  for (int k = threadIdx.x; k < K; k += 256) dr[k] = (row[row_off(g, k)] - cu * v[k]) / sg;
}

__global__ __launch_bounds__(256) void k_sn_bwd_dw(SnGeom g, const float* __restrict__ dweff,
                                                    const float* __restrict__ u, const float* __restrict__ v,
                                                    const float* __restrict__ sigma, const double* __restrict__ partial,
                                                    int K, float* __restrict__ dw) {
  extern __shared__ float row[];
  __shared__ double sm[16];
  d_sn_bwd_dw(g, dweff, u, v, sigma, partial, K, dw, blockIdx.x, sm, row);
}

// ---- multi-tensor forms: every spectrally normalised weight of a network in ONE launch per stage (blockIdx.z = tensor).
// A generator forward calls the hook on 18 convolutions and a PatchGAN pass on 3 per scale; one by one that is 4 launches
// of a few microseconds each per weight — ~200 launches per training step that individually cannot fill the chip.
#define SN_MAXT 12
struct SnFwdItem {
  const float* w;
  float *u, *v, *w_eff, *sigma, *u_used, *v_used, *part, *t, *sv;
  int Cout, K, R, cl_Cin, khw, scale_grid;
  long long n4;
};
struct SnFwdMulti {
  SnFwdItem it[SN_MAXT];
};
struct SnBwdItem {
  SnGeom g;
  const float *dweff, *w, *u, *v, *sigma;
  double* partial;
  float* dw;
  int K;
};
struct SnBwdMulti {
  SnBwdItem it[SN_MAXT];
};

__global__ __launch_bounds__(256) void k_sn_wtu_partial_m(SnFwdMulti m) {
  const SnFwdItem& a = m.it[blockIdx.z];
  if ((int)blockIdx.y >= a.R) return;
  d_sn_wtu_partial(a.w, a.u, a.Cout, a.K, a.R, a.part, blockIdx.x, blockIdx.y);
}
__global__ void k_sn_wtu_reduce_m(SnFwdMulti m) {
  const SnFwdItem& a = m.it[blockIdx.z];
  d_sn_wtu_reduce(a.part, a.K, a.R, a.t, blockIdx.x);
}
__global__ __launch_bounds__(256) void k_sn_rowdot_m(SnFwdMulti m, float eps, int iterate) {
  __shared__ double sm[16];
  const SnFwdItem& a = m.it[blockIdx.z];
  if ((int)blockIdx.x >= a.Cout) return;
  d_sn_rowdot(a.w, a.t, a.Cout, a.K, eps, iterate, a.v, a.v_used, a.sv, blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void k_sn_scale_m(SnFwdMulti m, float eps, int iterate) {
  __shared__ double sm[16];
  extern __shared__ __attribute__((aligned(16))) float row[];
  const SnFwdItem& a = m.it[blockIdx.z];
  if ((int)blockIdx.x >= a.scale_grid) return;
  d_sn_scale(a.w, a.sv, a.Cout, eps, iterate, a.u, a.u_used, a.sigma, a.n4, a.w_eff, a.cl_Cin, a.khw, blockIdx.x, a.scale_grid,
             sm, row);
}
__global__ __launch_bounds__(256) void k_sn_bwd_dot_m(SnBwdMulti m) {
  extern __shared__ float row[];
  __shared__ double sm[16];
  const SnBwdItem& a = m.it[blockIdx.z];
  if ((int)blockIdx.x >= a.g.Cout) return;
  d_sn_bwd_dot(a.g, a.dweff, a.w, a.K, a.partial, blockIdx.x, sm, row);
}
__global__ __launch_bounds__(256) void k_sn_bwd_dw_m(SnBwdMulti m) {
  extern __shared__ float row[];
  __shared__ double sm[16];
  const SnBwdItem& a = m.it[blockIdx.z];
  if ((int)blockIdx.x >= a.g.Cout) return;
  d_sn_bwd_dw(a.g, a.dweff, a.u, a.v, a.sigma, a.partial, a.K, a.dw, blockIdx.x, sm, row);
}

inline int sn_R(int64_t Cout, int64_t K) {
  const int64_t kb = cdiv(K, 1024);
  int64_t R = 512 / kb;
  if (R > Cout) R = Cout;
  if (R > 32) R = 32;                      // the second stage sums R rows per thread
  if (R < 1) R = 1;
  return (int)R;
}

}  // namespace

extern "C" {

int64_t csg_spectral_norm_workspace(int64_t Cout, int64_t K) {
  if (Cout <= 0 || K <= 0 || K % 4) return -1;
  const int64_t fwd = ((int64_t)sn_R(Cout, K) * K + K + Cout) * (int64_t)sizeof(float);
  const int64_t bwd = Cout * (int64_t)sizeof(double);
  return fwd > bwd ? fwd : bwd;
}

int csg_spectral_norm_fwd(const float* w, float* u, float* v, int64_t Cout, int64_t K, int iterate, float eps,
                          float* w_eff, int64_t cl_Cin, float* sigma, float* u_used, float* v_used, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  CSG_REQUIRE(cl_Cin == 0 || (cl_Cin % 4 == 0 && K % cl_Cin == 0), CSG_E_BADSHAPE,
              "csg_spectral_norm_fwd: channels-last output needs Cin %% 4 == 0 dividing K");
  CSG_REQUIRE(Cout > 0 && K > 0 && K % 4 == 0, CSG_E_BADSHAPE,
              "csg_spectral_norm_fwd: bad shape Cout=%ld K=%ld (K must be a multiple of 4)", (long)Cout, (long)K);
  CSG_REQUIRE(workspace && workspace_bytes >= csg_spectral_norm_workspace(Cout, K), CSG_E_BADSHAPE,
              "csg_spectral_norm_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_SPECTRAL_FWD, (double)Cout * K * 4 * (iterate ? 4 : 3), s);
  const int R = sn_R(Cout, K);
  float* part = (float*)workspace;
  float* t = part + (int64_t)R * K;
  float* sv = t + K;
  if (iterate) {
    CSG_LAUNCH(k_sn_wtu_partial, dim3((unsigned)cdiv(K, 1024), (unsigned)R), dim3(256), 0, s, w, u, (int)Cout,
                       (int)K, R, part);
    CSG_LAUNCH(k_sn_wtu_reduce, dim3((unsigned)cdiv(K, 256)), dim3(64), 0, s, part, (int)K, R, t);
  }
  CSG_LAUNCH(k_sn_rowdot, dim3((unsigned)Cout), dim3(256), 0, s, w, t, (int)Cout, (int)K, eps, iterate,
                     v, v_used, sv);
  const int64_t n4 = Cout * K / 4;
  int64_t grid = cdiv(n4, 256 * 4);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  size_t shm = 0;
  if (cl_Cin) {
    const int khw = (int)(K / cl_Cin);
    shm = (size_t)khw * (cl_Cin + 4) * sizeof(float);
    CSG_REQUIRE(shm <= 64 * 1024, CSG_E_UNSUPPORTED, "csg_spectral_norm_fwd: a %ld-element row does not fit the transpose buffer",
                (long)K);
    grid = Cout < 2048 ? Cout : 2048;
  }
  CSG_LAUNCH(k_sn_scale, dim3((unsigned)grid), dim3(256), shm, s, w, sv, (int)Cout, eps, iterate, u, u_used, sigma,
                     n4, w_eff, (int)cl_Cin, cl_Cin ? (int)(K / cl_Cin) : 1);
  return check_launch("csg_spectral_norm_fwd");
}

int csg_spectral_norm_bwd(const float* dweff, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t s0,
                          int64_t s1, int64_t s2, int64_t s3, const float* w, const float* u_used,
                          const float* v_used, const float* sigma, float* dw, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  CSG_REQUIRE(Cout > 0 && Cin > 0 && KH > 0 && KW > 0, CSG_E_BADSHAPE, "csg_spectral_norm_bwd: bad shape");
  const int64_t K = Cin * KH * KW;
  CSG_REQUIRE(K % 4 == 0 && K <= 15360, CSG_E_UNSUPPORTED,
              "csg_spectral_norm_bwd: K=%ld must be a multiple of 4 and at most 15360 (one LDS row)", (long)K);
  // rows of dW_eff must be dense: s0 == K and (s1,s2,s3) a dense permutation of (Cin,KH,KW)
  {
    int64_t d[3] = {Cin, KH, KW}, st[3] = {s1, s2, s3};
    for (int a = 0; a < 3; ++a)
      for (int c = a + 1; c < 3; ++c)
        if (st[c] < st[a]) {
          int64_t t = st[a]; st[a] = st[c]; st[c] = t;
          t = d[a]; d[a] = d[c]; d[c] = t;
        }
    int64_t run = 1;
    bool dense = s0 == K;
    for (int a = 0; a < 3; ++a) {
      if (d[a] > 1 && st[a] != run) dense = false;
      run *= d[a];
    }
    CSG_REQUIRE(dense, CSG_E_BADSHAPE, "csg_spectral_norm_bwd: dW_eff rows must be dense (strides %ld %ld %ld %ld)",
                (long)s0, (long)s1, (long)s2, (long)s3);
  }
  CSG_REQUIRE(workspace && workspace_bytes >= Cout * (int64_t)sizeof(double), CSG_E_BADSHAPE,
              "csg_spectral_norm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_SPECTRAL_BWD, (double)Cout * K * 4 * 4, s);
  SnGeom g{(int)Cout, (int)Cin, (int)KH, (int)KW, s0, s1, s2, s3};
  const size_t shm = (size_t)K * sizeof(float);
  CSG_LAUNCH(k_sn_bwd_dot, dim3((unsigned)Cout), dim3(256), shm, s, g, dweff, w, (int)K, (double*)workspace);
  CSG_LAUNCH(k_sn_bwd_dw, dim3((unsigned)Cout), dim3(256), shm, s, g, dweff, u_used, v_used, sigma,
                     (const double*)workspace, (int)K, dw);
  return check_launch("csg_spectral_norm_bwd");
}

// ---- multi-tensor entry points (see SnFwdMulti above): the same arithmetic, kernel by kernel, for up to SN_MAXT weights
// per launch (longer lists go out in chunks); every weight's results are bit-identical to the single-tensor calls.
int csg_spectral_norm_fwd_multi(const csg_sn_fwd_item* items, int32_t n, int32_t iterate, float eps, void* stream) {
  CSG_REQUIRE(items != nullptr && n > 0, CSG_E_BADSHAPE, "csg_spectral_norm_fwd_multi: empty list");
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += SN_MAXT) {
    const int cnt = n - base < SN_MAXT ? n - base : SN_MAXT;
    SnFwdMulti m;
    unsigned g_part_x = 1, g_part_y = 1, g_red = 1, g_row = 1, g_scale = 1;
    size_t shm = 0;
    double bytes = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const csg_sn_fwd_item& a = items[base + i];
      CSG_REQUIRE(a.cl_Cin == 0 || (a.cl_Cin % 4 == 0 && a.K % a.cl_Cin == 0), CSG_E_BADSHAPE,
                  "csg_spectral_norm_fwd_multi: channels-last output needs Cin %% 4 == 0 dividing K");
      CSG_REQUIRE(a.Cout > 0 && a.K > 0 && a.K % 4 == 0, CSG_E_BADSHAPE,
                  "csg_spectral_norm_fwd_multi: bad shape Cout=%ld K=%ld (K must be a multiple of 4)", (long)a.Cout, (long)a.K);
      CSG_REQUIRE(a.workspace && a.workspace_bytes >= csg_spectral_norm_workspace(a.Cout, a.K), CSG_E_BADSHAPE,
                  "csg_spectral_norm_fwd_multi: workspace too small");
      SnFwdItem& t = m.it[i];
      t.w = a.w; t.u = a.u; t.v = a.v; t.w_eff = a.w_eff; t.sigma = a.sigma; t.u_used = a.u_used; t.v_used = a.v_used;
      t.Cout = (int)a.Cout; t.K = (int)a.K; t.R = sn_R(a.Cout, a.K);
      t.part = (float*)a.workspace;
      t.t = t.part + (int64_t)t.R * a.K;
      t.sv = t.t + a.K;
      t.cl_Cin = (int)a.cl_Cin;
      t.khw = a.cl_Cin ? (int)(a.K / a.cl_Cin) : 1;
      t.n4 = a.Cout * a.K / 4;
      int64_t grid = cdiv(t.n4, 256 * 4);
      if (grid > 2048) grid = 2048;
      if (grid < 1) grid = 1;
      if (a.cl_Cin) {
        const size_t need = (size_t)t.khw * (a.cl_Cin + 4) * sizeof(float);
        CSG_REQUIRE(need <= 64 * 1024, CSG_E_UNSUPPORTED,
                    "csg_spectral_norm_fwd_multi: a %ld-element row does not fit the transpose buffer", (long)a.K);
        if (need > shm) shm = need;
        grid = a.Cout < 2048 ? a.Cout : 2048;
      }
      t.scale_grid = (int)grid;
      if ((unsigned)cdiv(a.K, 1024) > g_part_x) g_part_x = (unsigned)cdiv(a.K, 1024);
      if ((unsigned)t.R > g_part_y) g_part_y = (unsigned)t.R;
      if ((unsigned)cdiv(a.K, 256) > g_red) g_red = (unsigned)cdiv(a.K, 256);
      if ((unsigned)a.Cout > g_row) g_row = (unsigned)a.Cout;
      if ((unsigned)grid > g_scale) g_scale = (unsigned)grid;
      bytes += (double)a.Cout * a.K * 4 * (iterate ? 4 : 3);
    }
    ProfScope p(K_SPECTRAL_FWD, bytes, s);
    if (iterate) {
      CSG_LAUNCH(k_sn_wtu_partial_m, dim3(g_part_x, g_part_y, (unsigned)cnt), dim3(256), 0, s, m);
      CSG_LAUNCH(k_sn_wtu_reduce_m, dim3(g_red, 1, (unsigned)cnt), dim3(64), 0, s, m);
    }
    CSG_LAUNCH(k_sn_rowdot_m, dim3(g_row, 1, (unsigned)cnt), dim3(256), 0, s, m, eps, (int)iterate);
    CSG_LAUNCH(k_sn_scale_m, dim3(g_scale, 1, (unsigned)cnt), dim3(256), shm, s, m, eps, (int)iterate);
    const int rc = check_launch("csg_spectral_norm_fwd_multi");
    if (rc != CSG_OK) return rc;
  }
  return CSG_OK;
}

int csg_spectral_norm_bwd_multi(const csg_sn_bwd_item* items, int32_t n, void* stream) {
  CSG_REQUIRE(items != nullptr && n > 0, CSG_E_BADSHAPE, "csg_spectral_norm_bwd_multi: empty list");
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < n; base += SN_MAXT) {
    const int cnt = n - base < SN_MAXT ? n - base : SN_MAXT;
    SnBwdMulti m;
    unsigned g_row = 1;
    size_t shm = 0;
    double bytes = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const csg_sn_bwd_item& a = items[base + i];
      CSG_REQUIRE(a.Cout > 0 && a.Cin > 0 && a.KH > 0 && a.KW > 0, CSG_E_BADSHAPE, "csg_spectral_norm_bwd_multi: bad shape");
      const int64_t K = a.Cin * a.KH * a.KW;
      CSG_REQUIRE(K % 4 == 0 && K <= 15360, CSG_E_UNSUPPORTED,
                  "csg_spectral_norm_bwd_multi: K=%ld must be a multiple of 4 and at most 15360 (one LDS row)", (long)K);
      {
        int64_t d[3] = {a.Cin, a.KH, a.KW}, st[3] = {a.s1, a.s2, a.s3};
        for (int x = 0; x < 3; ++x)
          for (int c = x + 1; c < 3; ++c)
            if (st[c] < st[x]) {
              int64_t t = st[x]; st[x] = st[c]; st[c] = t;
              t = d[x]; d[x] = d[c]; d[c] = t;
            }
        int64_t run = 1;
        bool dense = a.s0 == K;
        for (int x = 0; x < 3; ++x) {
          if (d[x] > 1 && st[x] != run) dense = false;
          run *= d[x];
        }
        CSG_REQUIRE(dense, CSG_E_BADSHAPE, "csg_spectral_norm_bwd_multi: dW_eff rows must be dense (strides %ld %ld %ld %ld)",
                    (long)a.s0, (long)a.s1, (long)a.s2, (long)a.s3);
      }
      CSG_REQUIRE(a.workspace && a.workspace_bytes >= a.Cout * (int64_t)sizeof(double), CSG_E_BADSHAPE,
                  "csg_spectral_norm_bwd_multi: workspace too small");
      SnBwdItem& t = m.it[i];
      t.g = SnGeom{(int)a.Cout, (int)a.Cin, (int)a.KH, (int)a.KW, a.s0, a.s1, a.s2, a.s3};
      t.dweff = a.dweff; t.w = a.w; t.u = a.u_used; t.v = a.v_used; t.sigma = a.sigma;
      t.partial = (double*)a.workspace;
      t.dw = a.dw;
      t.K = (int)K;
      if ((unsigned)a.Cout > g_row) g_row = (unsigned)a.Cout;
      if ((size_t)K * sizeof(float) > shm) shm = (size_t)K * sizeof(float);
      bytes += (double)a.Cout * K * 4 * 4;
    }
    ProfScope p(K_SPECTRAL_BWD, bytes, s);
    CSG_LAUNCH(k_sn_bwd_dot_m, dim3(g_row, 1, (unsigned)cnt), dim3(256), shm, s, m);
    CSG_LAUNCH(k_sn_bwd_dw_m, dim3(g_row, 1, (unsigned)cnt), dim3(256), shm, s, m);
    const int rc = check_launch("csg_spectral_norm_bwd_multi");
    if (rc != CSG_OK) return rc;
  }
  return CSG_OK;
}

}  // extern "C"
