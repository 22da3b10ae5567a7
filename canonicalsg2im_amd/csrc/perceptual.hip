// Perceptual (VGG feature matching) loss helpers — the pieces of VGGLoss that are not convolutions:
// MaxPool2d(2,2) of the VGG19 feature stack and the L1 distance between two feature maps.
// HBM-bound streaming kernels over NHWC fp32, 16-byte accesses per lane; the L1 reduction is two
// stage (per-block fp64 partial, one block combines them in a fixed order) so it is deterministic.
//
// Reference: spade/models/networks/architecture.py:93-123 (VGG19 slices of torchvision's
// vgg19().features: conv3x3+ReLU blocks separated by MaxPool2d(kernel 2, stride 2)),
// spade/models/networks/loss.py:102-117 (VGGLoss: sum_i w_i * L1Loss(x_vgg[i], y_vgg[i].detach())).
#include "csg_common.h"

using namespace csg;

__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(float* p, float4 v) { *(float4*)p = v; }

// ---- MaxPool2d(2,2), floor mode: y[b,i,j,c] = max over x[b,2i+{0,1},2j+{0,1},c] ----------------
__global__ void k_maxpool2_fwd(const float* __restrict__ x, int H, int W, int OH, int OW, int Q, int64_t n4,
                               float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int ox = (int)(pix % OW);
    int64_t t = pix / OW;
    int oy = (int)(t % OH);
    int64_t b = t / OH;
    const float* p = x + (((b * H + 2 * oy) * W + 2 * ox) * (int64_t)Q + q) * 4;
    const int64_t row = (int64_t)W * Q * 4;
    float4 a = ld4(p), c = ld4(p + Q * 4), d = ld4(p + row), f = ld4(p + row + Q * 4);
    // NaN propagates as in ATen (val > max || isnan(val))
    auto mx = [](float m, float v) { return (v > m || v != v) ? v : m; };
    float4 r;
    r.x = mx(mx(mx(a.x, c.x), d.x), f.x);
    r.y = mx(mx(mx(a.y, c.y), d.y), f.y);
    r.z = mx(mx(mx(a.z, c.z), d.z), f.z);
    r.w = mx(mx(mx(a.w, c.w), d.w), f.w);
    st4(y + e * 4, r);
  }
}

// dx[b,h,w,c] = dy[b,h/2,w/2,c] if (h,w) is the FIRST maximum of its window in row-major order
// (ATen's max_pool2d keeps the first index on ties), else 0; rows/cols not covered by a window get 0.
__global__ void k_maxpool2_bwd(const float* __restrict__ dy, const float* __restrict__ x, int H, int W, int OH,
                               int OW, int Q, int64_t n4, float* __restrict__ dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int w = (int)(pix % W);
    int64_t t = pix / W;
    int h = (int)(t % H);
    int64_t b = t / H;
    int oy = h >> 1, ox = w >> 1;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (oy < OH && ox < OW) {
      const float* p = x + (((b * H + 2 * oy) * W + 2 * ox) * (int64_t)Q + q) * 4;
      const int64_t row = (int64_t)W * Q * 4;
      float4 v[4] = {ld4(p), ld4(p + Q * 4), ld4(p + row), ld4(p + row + Q * 4)};
      float4 g = ld4(dy + (((b * OH + oy) * OW + ox) * (int64_t)Q + q) * 4);
      const int me = ((h & 1) << 1) | (w & 1);
      auto arg = [](float a, float c, float d, float f) {
        int k = 0;
        float m = a;
        if (c > m || c != c) { m = c; k = 1; }
        if (d > m || d != d) { m = d; k = 2; }
        if (f > m || f != f) { m = f; k = 3; }
        return k;
      };
      r.x = arg(v[0].x, v[1].x, v[2].x, v[3].x) == me ? g.x : 0.f;
      r.y = arg(v[0].y, v[1].y, v[2].y, v[3].y) == me ? g.y : 0.f;
      r.z = arg(v[0].z, v[1].z, v[2].z, v[3].z) == me ? g.z : 0.f;
      r.w = arg(v[0].w, v[1].w, v[2].w, v[3].w) == me ? g.w : 0.f;
    }
    st4(dx + e * 4, r);
  }
}

// ---- mean |a - b| ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_l1_partial(const float* __restrict__ a, const float* __restrict__ b, int64_t n4,
                                                     double* __restrict__ partial) {
  __shared__ double red[4];
  float acc = 0.f;
  double tot = 0.0;
  int it = 0;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    float4 u = ld4(a + e * 4), v = ld4(b + e * 4);
    acc += (fabsf(u.x - v.x) + fabsf(u.y - v.y)) + (fabsf(u.z - v.z) + fabsf(u.w - v.w));
    if (++it == 64) {  // flush the fp32 running sum before it grows long
      tot += (double)acc;
      acc = 0.f;
      it = 0;
    }
  }
  tot += (double)acc;
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_l1_final(const double* __restrict__ partial, int nblk, double inv_n,
                                                   float* __restrict__ out) {
  __shared__ double red[4];
  double tot = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) tot += partial[i];
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) red[wv] = tot;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * inv_n);
}

// da = sign(a - b) * g / n   (sign(0) = 0, as torch's l1_loss backward)
__global__ void k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                         float inv_n, int64_t n4, float* __restrict__ da) {
  const float s = g[0] * inv_n;
  auto sg = [s](float d) { return d > 0.f ? s : (d < 0.f ? -s : 0.f); };
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    float4 u = ld4(a + e * 4), v = ld4(b + e * 4);
    st4(da + e * 4, make_float4(sg(u.x - v.x), sg(u.y - v.y), sg(u.z - v.z), sg(u.w - v.w)));
  }
}

static inline unsigned ew_grid(int64_t n) {
  int64_t g = cdiv(n, 256);
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

// ---- AvgPool2d(2,2), floor mode (sg2im/layers.py:88-90, `build_cnn(pooling='avg')`): mean of the 2 x 2 window; backward
// hands each covered input pixel a quarter of its window's gradient, rows / columns no window covers get 0
__global__ void k_avgpool2_fwd(const float* __restrict__ x, int H, int W, int OH, int OW, int Q, int64_t n4,
                               float* __restrict__ y) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int ox = (int)(pix % OW);
    int64_t t = pix / OW;
    int oy = (int)(t % OH);
    int64_t b = t / OH;
    const float* p = x + (((b * H + 2 * oy) * W + 2 * ox) * (int64_t)Q + q) * 4;
    const int64_t row = (int64_t)W * Q * 4;
    const float4 a = ld4(p), c = ld4(p + Q * 4), d = ld4(p + row), f = ld4(p + row + Q * 4);
    float4 r;                                  // ATen's order: the window summed row by row, then divided by its size
    r.x = (((a.x + c.x) + d.x) + f.x) / 4.f;
    r.y = (((a.y + c.y) + d.y) + f.y) / 4.f;
    r.z = (((a.z + c.z) + d.z) + f.z) / 4.f;
    r.w = (((a.w + c.w) + d.w) + f.w) / 4.f;
    st4(y + e * 4, r);
  }
}

__global__ void k_avgpool2_bwd(const float* __restrict__ dy, int H, int W, int OH, int OW, int Q, int64_t n4,
                               float* __restrict__ dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (int64_t)gridDim.x * blockDim.x) {
    int64_t pix = e / Q;
    int q = (int)(e - pix * Q);
    int w = (int)(pix % W);
    int64_t t = pix / W;
    int h = (int)(t % H);
    int64_t b = t / H;
    const int oy = h >> 1, ox = w >> 1;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (oy < OH && ox < OW) {
      const float4 g = ld4(dy + (((b * OH + oy) * OW + ox) * (int64_t)Q + q) * 4);
      r = make_float4(g.x / 4.f, g.y / 4.f, g.z / 4.f, g.w / 4.f);
    }
    st4(dx + e * 4, r);
  }
}

// ---- hinge / Wasserstein-style GAN terms on the PatchGAN's prediction maps (reference spade/models/networks/loss.py:60-93) ----
// kind 0: -mean(x) (generator), 1: -mean(min(x - 1, 0)) (discriminator, real), 2: -mean(min(-x - 1, 0)) (discriminator, fake),
// averaged over up to four scales: ONE launch where torch ran sub / clamp / mean / neg per scale + the sum and the division
// (and as many in the backward).  A map is (B, 1, h, w) with arbitrary element strides (the one real channel of a padded
// NHWC buffer).  fp64 sums in a fixed order (one block, strided per thread, LDS tree): bit-reproducible.
struct HingeItems {
  const float* x[4];
  float* dx[4];                 // backward: contiguous (B, h, w)
  long long sb[4], sh[4], sw[4];
  int B[4], H[4], W[4];
  int n;
};
__device__ __forceinline__ float hinge_term(float x, int kind) {
  if (kind == 0) return x;
  const float m = kind == 1 ? x - 1.0f : -x - 1.0f;
  return m < 0.f ? m : 0.f;
}
__global__ __launch_bounds__(1024) void k_hinge_mean(HingeItems it, int kind, float* __restrict__ out) {
  __shared__ double red[1024];
  double total = 0.0;
  for (int i = 0; i < it.n; ++i) {
    const int hw = it.H[i] * it.W[i];
    const int N = it.B[i] * hw;
    double acc = 0.0;
    for (int e = threadIdx.x; e < N; e += 1024) {
      const int b = e / hw, r = e - b * hw, y = r / it.W[i], xx = r - y * it.W[i];
      acc += (double)hinge_term(it.x[i][b * it.sb[i] + y * it.sh[i] + xx * it.sw[i]], kind);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    total += -(red[0] / (double)N);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(total / (double)it.n);
}
// d x = g * (-1 / (N n)) * d term / d x
__global__ void k_hinge_bwd(HingeItems it, int kind, const float* __restrict__ g) {
  const int i = blockIdx.y;
  const int hw = it.H[i] * it.W[i];
  const int N = it.B[i] * hw;
  const float s = -g[0] / ((float)N * (float)it.n);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < N; e += gridDim.x * blockDim.x) {
    const int b = e / hw, r = e - b * hw, y = r / it.W[i], xx = r - y * it.W[i];
    const float x = it.x[i][b * it.sb[i] + y * it.sh[i] + xx * it.sw[i]];
    float d = 1.0f;
    if (kind == 1) d = (x - 1.0f) < 0.f ? 1.0f : 0.f;
    if (kind == 2) d = (-x - 1.0f) < 0.f ? -1.0f : 0.f;
    it.dx[i][e] = s * d;
  }
}

extern "C" {

int csg_maxpool2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream) {
  CSG_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_maxpool2_fwd: bad shape B=%ld H=%ld W=%ld C=%ld", (long)B, (long)H, (long)W, (long)C);
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t n4 = B * OH * OW * C / 4;
  ProfScope p(K_MAXPOOL_FWD, (double)(B * OH * OW * C) * 5 * 4, s);
  CSG_LAUNCH(k_maxpool2_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, (int)H, (int)W, (int)OH, (int)OW,
                     (int)(C / 4), n4, y);
  return check_launch("csg_maxpool2_fwd");
}

int csg_maxpool2_bwd(const float* dy, const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* dx,
                     void* stream) {
  CSG_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_maxpool2_bwd: bad shape B=%ld H=%ld W=%ld C=%ld", (long)B, (long)H, (long)W, (long)C);
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t n4 = B * H * W * C / 4;
  ProfScope p(K_MAXPOOL_BWD, (double)(B * H * W * C) * 2.25 * 4, s);
  CSG_LAUNCH(k_maxpool2_bwd, dim3(ew_grid(n4)), dim3(256), 0, s, dy, x, (int)H, (int)W, (int)OH, (int)OW,
                     (int)(C / 4), n4, dx);
  return check_launch("csg_maxpool2_bwd");
}

int csg_avgpool2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream) {
  CSG_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_avgpool2_fwd: bad shape B=%ld H=%ld W=%ld C=%ld", (long)B, (long)H, (long)W, (long)C);
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t n4 = B * OH * OW * C / 4;
  ProfScope p(K_AVGPOOL_FWD, (double)(B * OH * OW * C) * 5 * 4, s);
  CSG_LAUNCH(k_avgpool2_fwd, dim3(ew_grid(n4)), dim3(256), 0, s, x, (int)H, (int)W, (int)OH, (int)OW, (int)(C / 4), n4, y);
  return check_launch("csg_avgpool2_fwd");
}

int csg_avgpool2_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream) {
  CSG_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, CSG_E_BADSHAPE,
              "csg_avgpool2_bwd: bad shape B=%ld H=%ld W=%ld C=%ld", (long)B, (long)H, (long)W, (long)C);
  hipStream_t s = (hipStream_t)stream;
  const int64_t OH = H / 2, OW = W / 2;
  const int64_t n4 = B * H * W * C / 4;
  ProfScope p(K_AVGPOOL_BWD, (double)(B * H * W * C) * 1.25 * 4, s);
  CSG_LAUNCH(k_avgpool2_bwd, dim3(ew_grid(n4)), dim3(256), 0, s, dy, (int)H, (int)W, (int)OH, (int)OW, (int)(C / 4), n4, dx);
  return check_launch("csg_avgpool2_bwd");
}

int64_t csg_l1_mean_workspace(int64_t n) {
  if (n <= 0 || n % 4) return -1;
  return (int64_t)ew_grid(cdiv(n / 4, 4)) * (int64_t)sizeof(double);
}

int csg_l1_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                    void* stream) {
  CSG_REQUIRE(n > 0 && n % 4 == 0, CSG_E_BADSHAPE, "csg_l1_mean_fwd: n=%ld must be a positive multiple of 4", (long)n);
  const unsigned nblk = ew_grid(cdiv(n / 4, 4));
  CSG_REQUIRE(workspace && workspace_bytes >= (int64_t)nblk * (int64_t)sizeof(double), CSG_E_BADSHAPE,
              "csg_l1_mean_fwd: workspace too small (%ld bytes, need %ld)", (long)workspace_bytes,
              (long)(nblk * sizeof(double)));
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_L1_FWD, (double)n * 2 * 4, s);
  CSG_LAUNCH(k_l1_partial, dim3(nblk), dim3(256), 0, s, a, b, n / 4, (double*)workspace);
  CSG_LAUNCH(k_l1_final, dim3(1), dim3(256), 0, s, (const double*)workspace, (int)nblk, 1.0 / (double)n, out);
  return check_launch("csg_l1_mean_fwd");
}

int csg_l1_mean_bwd(const float* a, const float* b, const float* gout, int64_t n, float* da, void* stream) {
  CSG_REQUIRE(n > 0 && n % 4 == 0, CSG_E_BADSHAPE, "csg_l1_mean_bwd: n=%ld must be a positive multiple of 4", (long)n);
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_L1_BWD, (double)n * 3 * 4, s);
  CSG_LAUNCH(k_l1_bwd, dim3(ew_grid(n / 4)), dim3(256), 0, s, a, b, gout, (float)(1.0 / (double)n), n / 4, da);
  return check_launch("csg_l1_mean_bwd");
}

static int hinge_items(const csg_hinge_item* items, int32_t n, bool need_dx, HingeItems& it, const char* who) {
  CSG_REQUIRE(items != nullptr && n >= 1 && n <= 4, CSG_E_BADSHAPE, "%s: 1..4 maps (got %d)", who, (int)n);
  it.n = n;
  for (int i = 0; i < 4; ++i) {
    const csg_hinge_item& d = items[i < n ? i : 0];
    CSG_REQUIRE(d.x != nullptr && d.B > 0 && d.H > 0 && d.W > 0 && d.B * d.H * d.W < (1ll << 30) && (!need_dx || d.dx != nullptr),
                CSG_E_BADSHAPE, "%s: bad map %d", who, i);
    it.x[i] = d.x; it.dx[i] = d.dx; it.sb[i] = d.sb; it.sh[i] = d.sh; it.sw[i] = d.sw;
    it.B[i] = (int)d.B; it.H[i] = (int)d.H; it.W[i] = (int)d.W;
  }
  return CSG_OK;
}

int csg_hinge_mean_fwd(const csg_hinge_item* items, int32_t n, int32_t kind, float* out, void* stream) {
  CSG_REQUIRE(kind >= 0 && kind <= 2 && out != nullptr, CSG_E_BADSHAPE, "csg_hinge_mean_fwd: kind 0..2, an output");
  HingeItems it;
  int rc = hinge_items(items, n, false, it, "csg_hinge_mean_fwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  double bytes = 0;
  for (int i = 0; i < n; ++i) bytes += (double)it.B[i] * it.H[i] * it.W[i] * 4;
  ProfScope p(K_L1_FWD, bytes, s);
  CSG_LAUNCH(k_hinge_mean, dim3(1), dim3(1024), 0, s, it, (int)kind, out);
  return check_launch("csg_hinge_mean_fwd");
}

int csg_hinge_mean_bwd(const csg_hinge_item* items, int32_t n, int32_t kind, const float* gout, void* stream) {
  CSG_REQUIRE(kind >= 0 && kind <= 2 && gout != nullptr, CSG_E_BADSHAPE, "csg_hinge_mean_bwd: kind 0..2, a gradient");
  HingeItems it;
  int rc = hinge_items(items, n, true, it, "csg_hinge_mean_bwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  int64_t nmax = 0;
  for (int i = 0; i < n; ++i) nmax = nmax > (int64_t)it.B[i] * it.H[i] * it.W[i] ? nmax : (int64_t)it.B[i] * it.H[i] * it.W[i];
  ProfScope p(K_L1_BWD, (double)nmax * 8 * n, s);
  CSG_LAUNCH(k_hinge_bwd, dim3(ew_grid(nmax), (unsigned)n), dim3(256), 0, s, it, (int)kind, gout);
  return check_launch("csg_hinge_mean_bwd");
}

}  // extern "C"
