// Convolutions with at most four output channels (reference: `conv_img`, generator.py:46,120-121 — 64 -> 3 channels,
// 3x3 — and the PatchGAN prediction heads, discriminator.py:185-187 — 512 -> 1 channel, 4x4, padding 2), stride 1.
//
// On the matrix cores these waste a 32-wide tile on 3 (or 1) useful columns (7-30 TFLOP/s); their arithmetic is tiny
// and the work is moving the many-channel side once.  Three VALU kernels, NHWC fp32, output channels padded to 4:
//   forward        thread = output pixel; the input patch of a 8x32 tile is staged in LDS 16 channels at a time, the
//                  weights come in through scalar loads (wave-uniform addresses);
//   backward-data  thread = (input pixel, channel quad) with its KH*KW*n weight quads in registers; dX written once,
//                  coalesced;
//   weight grad    thread = (channel quad, pixel lane) walks INPUT pixels, so x is read exactly once; the
//                  KH*KW*n accumulator quads are summed over the pixel lanes through LDS in a fixed order, one slab per
//                  block, ordered slab sum (csg_reduce.h): bit-reproducible, no atomics.
#include "csg_common.h"
#include "csg_reduce.h"

using namespace csg;

struct FewParams {
  int B, IH, IW, Cin, x_cs, OH, OW, pad, nreal, act;
  float slope;
  int in_act;         // 1: the convolution sees leaky(x, in_slope) — the activation in front of it applied in the loaders
  float in_slope;
};

__device__ __forceinline__ float4 few_in_act(float4 v, float s) {
  return make_float4(v.x > 0.f ? v.x : v.x * s, v.y > 0.f ? v.y : v.y * s, v.z > 0.f ? v.z : v.z * s, v.w > 0.f ? v.w : v.w * s);
}

#define FW_CK 16
#define FW_LD 20

// Forward.  Block = 256 output pixels (TY x TX, TX = 32 or 16) x ONE range of 32-channel chunks (`cps` chunks per
// split: maps with few pixels and many channels — the PatchGAN heads — are cut along the channels so that the grid
// fills the chip; the partial sums go to slabs, k_few_finish adds them in order and applies bias / activation).
template <int KH, int KW, int NR>
__global__ __launch_bounds__(256) void k_few_fwd(FewParams p, const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, int TX, int ksplit,
                                                  int cps, int64_t slab) {
  const int TY = 256 / TX;
  const int PH = TY + KH - 1, PW = TX + KW - 1;
  extern __shared__ __attribute__((aligned(16))) float patch[];        // [PH][PW][FW_LD]
  const int tid = threadIdx.x, tx = tid % TX, ty = tid / TX;
  const int tbx = (p.OW + TX - 1) / TX, tby = (p.OH + TY - 1) / TY;
  int bid = blockIdx.x;
  const int split = bid % ksplit;
  bid /= ksplit;
  const int bx = bid % tbx;
  bid /= tbx;
  const int by = bid % tby;
  const int b = bid / tby;
  const int ox0 = bx * TX, oy0 = by * TY;
  float acc[NR];
#pragma unroll
  for (int n = 0; n < NR; ++n) acc[n] = 0.f;
  const int c_begin = split * cps * FW_CK;
  const int c_end = min(p.Cin, c_begin + cps * FW_CK);
  for (int c0 = c_begin; c0 < c_end; c0 += FW_CK) {
    // (19 x 19 or 11 x 35 pixels) x 4 quads <= 7 x 256: every load of a thread is issued before its first LDS store
    float4 st[7];
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int e = tid + 256 * it;
      const int pix = e >> 2, c4 = e & 3;
      const int r = pix / PW, c = pix - r * PW;
      const int iy = oy0 + r - p.pad, ix = ox0 + c - p.pad;
      st[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < PH * PW * (FW_CK / 4) && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW)
        st[it] = *(const float4*)(x + ((int64_t)(b * p.IH + iy) * p.IW + ix) * p.x_cs + c0 + c4 * 4);
    }
#pragma unroll
    for (int it = 0; it < 7; ++it) {
      const int e = tid + 256 * it;
      if (e < PH * PW * (FW_CK / 4)) *(float4*)(patch + (e >> 2) * FW_LD + (e & 3) * 4) = p.in_act ? few_in_act(st[it], p.in_slope) : st[it];
    }
    __syncthreads();
    // one tap at a time: 4 * NR scalar weight quads live at once (fully unrolled, the compiler hoists every weight of
    // the chunk into SGPRs and spills them); 16-channel chunks keep the patch at ~25 KB: six blocks per CU hide the
    // scalar-load latency
#pragma unroll 1
    for (int t = 0; t < KH * KW; ++t) {
      const int kh = t / KW, kw = t - kh * KW;
      const float* px = patch + ((ty + kh) * PW + tx + kw) * FW_LD;
      const float* pw = w + (int64_t)t * p.Cin + c0;                      // wave-uniform: scalar loads
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float4 xv = *(const float4*)(px + c4 * 4);
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          const float4 wv = *(const float4*)(pw + (int64_t)n * KH * KW * p.Cin + c4 * 4);
          acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
      }
    }
    __syncthreads();
  }
  const int oy = oy0 + ty, ox = ox0 + tx;
  if (oy < p.OH && ox < p.OW) {
    float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NR; ++n) out[n] = acc[n];
    const int64_t o = (int64_t)(b * p.OH + oy) * p.OW + ox;
    if (ksplit > 1) {
      *(float4*)(y + (int64_t)split * slab + o * 4) = make_float4(out[0], out[1], out[2], out[3]);
      return;
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      float v = out[n] + (bias != nullptr && n < p.nreal ? bias[n] : 0.f);
      if (p.act == CSG_ACT_LEAKY)
        v = v > 0.f ? v : v * p.slope;
      else if (p.act == CSG_ACT_TANH)
        v = tanhf(v);
      out[n] = v;
    }
    *(float4*)(y + o * 4) = make_float4(out[0], out[1], out[2], out[3]);
  }
}

__global__ void k_few_finish(FewParams p, const float4* __restrict__ slabs, int ksplit, int64_t npix,
                             const float* __restrict__ bias, float4* __restrict__ y) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= npix) return;
  float4 a = slabs[o];
  int s = 1;
  for (; s + 4 <= ksplit; s += 4) {            // four loads in flight, added in slab order
    float4 t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = slabs[(int64_t)(s + u) * npix + o];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
  }
  for (; s < ksplit; ++s) {
    const float4 v = slabs[(int64_t)s * npix + o];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float out[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    float v = out[n] + (bias != nullptr && n < p.nreal ? bias[n] : 0.f);
    if (p.act == CSG_ACT_LEAKY)
      v = v > 0.f ? v : v * p.slope;
    else if (p.act == CSG_ACT_TANH)
      v = tanhf(v);
    out[n] = v;
  }
  y[o] = make_float4(out[0], out[1], out[2], out[3]);
}

// Pixel walk of the two backward kernels: thread lane `pl` of PL visits q = q0 + pl, q0 + pl + PL, ... (< q1) with its
// (image, row, column) kept incrementally — no divisions in the loop.
struct PixWalk {
  int b, iy, ix;
  __device__ __forceinline__ void start(long long q, int IH, int IW) {
    ix = (int)(q % IW);
    const long long t = q / IW;
    iy = (int)(t % IH);
    b = (int)(t / IH);
  }
  __device__ __forceinline__ void advance(int step, int IH, int IW) {
    ix += step;
    while (ix >= IW) {
      ix -= IW;
      if (++iy == IH) {
        iy = 0;
        ++b;
      }
    }
  }
};

// dX[q][c] = sum_{tap, n} dY[q + pad - tap][n] * w[n][tap][c]
template <int KH, int KW, int NR>
__global__ __launch_bounds__(256) void k_few_bwd_data(FewParams p, const float* __restrict__ dy,
                                                       const float* __restrict__ w, float* __restrict__ dx, int64_t npix,
                                                       int ppb) {
  const int C4 = p.Cin >> 2, PL = 256 / C4;
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  float4 wr[KH * KW][NR];
#pragma unroll
  for (int t = 0; t < KH * KW; ++t)
#pragma unroll
    for (int n = 0; n < NR; ++n) wr[t][n] = *(const float4*)(w + ((int64_t)n * KH * KW + t) * p.Cin + c4 * 4);
  const int64_t q0 = (int64_t)blockIdx.x * ppb;
  const int64_t q1 = q0 + ppb < npix ? q0 + ppb : npix;
  if (q0 + pl >= q1) return;
  PixWalk pw;
  pw.start(q0 + pl, p.IH, p.IW);
  for (int64_t q = q0 + pl; q < q1; q += PL) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* dyb = dy + (int64_t)pw.b * p.OH * p.OW * 4;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int oy = pw.iy + p.pad - kh;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int ox = pw.ix + p.pad - kw;
        // branch-free: an out-of-range tap reads pixel 0 and is multiplied by zero
        const bool ok = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
        const float4 g = *(const float4*)(dyb + (ok ? (oy * p.OW + ox) * 4 : 0));
        const float m = ok ? 1.f : 0.f;
        const float gv[4] = {g.x * m, g.y * m, g.z * m, g.w * m};
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          const float4 wv = wr[kh * KW + kw][n];
          acc.x += gv[n] * wv.x; acc.y += gv[n] * wv.y; acc.z += gv[n] * wv.z; acc.w += gv[n] * wv.w;
        }
      }
    }
    *(float4*)(dx + q * p.x_cs + c4 * 4) = acc;
    pw.advance(PL, p.IH, p.IW);
  }
}

// dW[n][tap][c] = sum_q x[q][c] * dY[q + pad - tap][n]; one slab [4][KH*KW][Cin] (+ 4 bias sums) per block
#define FEW_RED 8      // accumulator quads summed over the pixel lanes per LDS round
template <int KH, int KW, int NR>
__global__ __launch_bounds__(256) void k_few_bwd_weight(FewParams p, const float* __restrict__ x,
                                                         const float* __restrict__ dy, float* __restrict__ slabs,
                                                         float* __restrict__ dbslabs, int64_t npix, int ppb,
                                                         int64_t nout, int opb) {
  constexpr int T = KH * KW;
  __shared__ float4 part[FEW_RED][256];                                // [slot][pl * C4 + c4]
  const int C4 = p.Cin >> 2, PL = 256 / C4;
  const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4;
  float4 acc[T * NR];
#pragma unroll
  for (int i = 0; i < T * NR; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t q0 = (int64_t)blockIdx.x * ppb;
  const int64_t q1 = q0 + ppb < npix ? q0 + ppb : npix;
  PixWalk pw;
  pw.start(q0 + pl < npix ? q0 + pl : 0, p.IH, p.IW);
  for (int64_t q = q0 + pl; q < q1; q += PL) {
    float4 xv = *(const float4*)(x + q * p.x_cs + c4 * 4);
    if (p.in_act) xv = few_in_act(xv, p.in_slope);
    const float* dyb = dy + (int64_t)pw.b * p.OH * p.OW * 4;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int oy = pw.iy + p.pad - kh;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int ox = pw.ix + p.pad - kw;
        const bool ok = (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
        const float4 g = *(const float4*)(dyb + (ok ? (oy * p.OW + ox) * 4 : 0));
        const float m = ok ? 1.f : 0.f;
        const float gv[4] = {g.x * m, g.y * m, g.z * m, g.w * m};
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          float4& a = acc[(kh * KW + kw) * NR + n];
          a.x += gv[n] * xv.x; a.y += gv[n] * xv.y; a.z += gv[n] * xv.z; a.w += gv[n] * xv.w;
        }
      }
    }
    pw.advance(PL, p.IH, p.IW);
  }
  // sum over the pixel lanes in lane order, FEW_RED accumulators per round; thread (pl, c4) finishes slot pl
  float* slab = slabs + (int64_t)blockIdx.x * 4 * T * p.Cin;
#pragma unroll
  for (int r0 = 0; r0 < T * NR; r0 += FEW_RED) {
#pragma unroll
    for (int j = 0; j < FEW_RED; ++j)
      if (r0 + j < T * NR) part[j][pl * C4 + c4] = acc[r0 + j];
    __syncthreads();
    for (int j = pl; j < FEW_RED && r0 + j < T * NR; j += PL) {
      float4 s = part[j][c4];
      for (int l = 1; l < PL; ++l) {
        const float4 v = part[j][l * C4 + c4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int i = r0 + j, t = i / NR, n = i - t * NR;
      *(float4*)(slab + ((int64_t)n * T + t) * p.Cin + c4 * 4) = s;
    }
    __syncthreads();
  }
  if (pl == 0) {
    for (int n = NR; n < 4; ++n)
      for (int t = 0; t < T; ++t) *(float4*)(slab + ((int64_t)n * T + t) * p.Cin + c4 * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // bias gradient: this block's share of the OUTPUT pixels, summed in a fixed order
  if (dbslabs != nullptr) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t o0 = (int64_t)blockIdx.x * opb;
    const int64_t o1 = o0 + opb < nout ? o0 + opb : nout;
    for (int64_t o = o0 + threadIdx.x; o < o1; o += 256) {
      const float4 g = *(const float4*)(dy + o * 4);
      s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    }
    part[0][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float4 tot = part[0][0];
      for (int l = 1; l < 256; ++l) {
        const float4 v = part[0][l];
        tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
      }
      *(float4*)(dbslabs + (int64_t)blockIdx.x * 4) = tot;
    }
  }
}

// ------------------------------------------------------------------------------------ host side
static int few_plan(const csg_few_desc* d, FewParams& p, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->IH > 0 && d->IW > 0 && d->Cin > 0 && d->pad >= 0, CSG_E_BADSHAPE, "%s: bad dimension", who);
  CSG_REQUIRE((d->KH == 3 && d->KW == 3) || (d->KH == 4 && d->KW == 4), CSG_E_UNSUPPORTED,
              "%s: only 3x3 and 4x4 kernels (conv_img, the PatchGAN heads), got %dx%d", who, d->KH, d->KW);
  CSG_REQUIRE(d->Cin % 32 == 0 && d->Cin <= 1024 && 1024 % d->Cin == 0 && d->x_cs % 4 == 0 && d->x_cs >= d->Cin,
              CSG_E_UNSUPPORTED, "%s: Cin=%d must be 32, 64, ..., 1024", who, d->Cin);
  CSG_REQUIRE(d->cout_real >= 1 && d->cout_real <= 4, CSG_E_UNSUPPORTED, "%s: 1..4 output channels", who);
  CSG_REQUIRE(d->KH * d->KW * d->cout_real <= 36, CSG_E_UNSUPPORTED, "%s: %dx%d taps x %d outputs do not fit the registers", who,
              d->KH, d->KW, d->cout_real);
  p.B = d->B; p.IH = d->IH; p.IW = d->IW; p.Cin = d->Cin; p.x_cs = d->x_cs; p.pad = d->pad;
  p.OH = d->IH + 2 * d->pad - d->KH + 1;
  p.OW = d->IW + 2 * d->pad - d->KW + 1;
  CSG_REQUIRE(p.OH > 0 && p.OW > 0, CSG_E_BADSHAPE, "%s: empty output", who);
  p.nreal = d->cout_real; p.act = d->act; p.slope = d->slope;
  p.in_act = d->in_act; p.in_slope = d->in_slope;
  CSG_REQUIRE((int64_t)d->B * d->IH * d->IW * d->x_cs < (1ll << 40), CSG_E_UNSUPPORTED, "%s: tensor too large", who);
  return CSG_OK;
}

// pixels per block for the pixel-walking kernels: ~1024 blocks, at least 32 pixels per pixel lane (the in-block
// reduction of the weight gradient is paid once per block)
static int few_ppb(int64_t npix, int Cin) {
  const int PL = 256 / (Cin / 4);
  int64_t ppb = (npix + 1023) / 1024;
  if (ppb < 32 * PL) ppb = 32 * PL;
  ppb = (ppb + PL - 1) / PL * PL;
  return (int)ppb;
}

struct FewFwdPlan {
  int TX, ksplit, cps;
  int64_t grid, slab;
  size_t shm;
};
static FewFwdPlan few_fwd_plan(const FewParams& p, int KH, int KW) {
  FewFwdPlan f;
  f.TX = p.OW > 48 ? 32 : 16;
  const int TY = 256 / f.TX;
  const int64_t tiles = (int64_t)p.B * cdiv(p.OH, TY) * cdiv(p.OW, f.TX);
  const int nchunk = p.Cin / FW_CK;
  int ks = 1;
  if (tiles < 1024) ks = (int)((1536 + tiles - 1) / tiles);
  if (ks > nchunk) ks = nchunk;
  f.cps = (nchunk + ks - 1) / ks;
  f.ksplit = (nchunk + f.cps - 1) / f.cps;
  f.grid = tiles * f.ksplit;
  f.slab = (int64_t)p.B * p.OH * p.OW * 4;
  f.shm = (size_t)(TY + KH - 1) * (f.TX + KW - 1) * FW_LD * 4;
  return f;
}

template <int KH, int KW, int NR>
static void few_launch_fwd(const FewParams& p, const FewFwdPlan& f, const float* x, const float* w, const float* bias,
                           float* out, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_few_fwd<KH, KW, NR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  CSG_LAUNCH((k_few_fwd<KH, KW, NR>), dim3((unsigned)f.grid), dim3(256), f.shm, s, p, x, w, bias, out, f.TX, f.ksplit,
                     f.cps, f.slab);
}

extern "C" {

int csg_conv_few_supported(const csg_few_desc* d) {
  FewParams p;
  const int rc = few_plan(d, p, "csg_conv_few_supported");
  return rc == CSG_OK ? 1 : 0;
}

int64_t csg_conv_few_fwd_workspace(const csg_few_desc* d) {
  FewParams p;
  if (few_plan(d, p, "csg_conv_few_fwd_workspace")) return -1;
  const FewFwdPlan f = few_fwd_plan(p, d->KH, d->KW);
  return f.ksplit > 1 ? f.ksplit * f.slab * 4 : 0;
}

int csg_conv_few_fwd(const csg_few_desc* d, const float* x, const float* w, const float* bias, float* y, float* workspace,
                     int64_t workspace_bytes, void* stream) {
  FewParams p;
  int rc = few_plan(d, p, "csg_conv_few_fwd");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)y % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_few_fwd: pointers must be 16-byte aligned");
  FewFwdPlan f = few_fwd_plan(p, d->KH, d->KW);
  if (f.ksplit > 1 && (workspace == nullptr || workspace_bytes < f.ksplit * f.slab * 4 || ((uintptr_t)workspace % 16) != 0)) {
    f.grid /= f.ksplit;                                   // no slabs: one block walks all channel chunks
    f.ksplit = 1;
    f.cps = p.Cin / FW_CK;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(K_FEW_FWD, ((double)p.B * p.IH * p.IW * p.Cin + (double)p.B * p.OH * p.OW * 4) * 4, s);
  float* out = f.ksplit > 1 ? workspace : y;
  if (d->KH == 3) {
    if (p.nreal == 1) few_launch_fwd<3, 3, 1>(p, f, x, w, bias, out, s);
    else if (p.nreal == 2) few_launch_fwd<3, 3, 2>(p, f, x, w, bias, out, s);
    else if (p.nreal == 3) few_launch_fwd<3, 3, 3>(p, f, x, w, bias, out, s);
    else few_launch_fwd<3, 3, 4>(p, f, x, w, bias, out, s);
  } else {
    if (p.nreal == 1) few_launch_fwd<4, 4, 1>(p, f, x, w, bias, out, s);
    else few_launch_fwd<4, 4, 2>(p, f, x, w, bias, out, s);
  }
  rc = check_launch("csg_conv_few_fwd");
  if (rc == CSG_OK && f.ksplit > 1) {
    const int64_t npix = (int64_t)p.B * p.OH * p.OW;
    CSG_LAUNCH(k_few_finish, dim3((unsigned)cdiv(npix, 256)), dim3(256), 0, s, p, (const float4*)workspace, f.ksplit,
                       npix, bias, (float4*)y);
    rc = check_launch("csg_conv_few_fwd(finish)");
  }
  return rc;
}

#define FEW_DISPATCH(KERNEL, ...)                                                                  \
  do {                                                                                             \
    if (d->KH == 3) {                                                                              \
      if (p.nreal == 1) CSG_LAUNCH((KERNEL<3, 3, 1>), __VA_ARGS__);                        \
      else if (p.nreal == 2) CSG_LAUNCH((KERNEL<3, 3, 2>), __VA_ARGS__);                   \
      else if (p.nreal == 3) CSG_LAUNCH((KERNEL<3, 3, 3>), __VA_ARGS__);                   \
      else CSG_LAUNCH((KERNEL<3, 3, 4>), __VA_ARGS__);                                     \
    } else {                                                                                       \
      if (p.nreal == 1) CSG_LAUNCH((KERNEL<4, 4, 1>), __VA_ARGS__);                        \
      else CSG_LAUNCH((KERNEL<4, 4, 2>), __VA_ARGS__);                                     \
    }                                                                                              \
  } while (0)

int csg_conv_few_bwd_data(const csg_few_desc* d, const float* dy, const float* w, float* dx, void* stream) {
  FewParams p;
  int rc = few_plan(d, p, "csg_conv_few_bwd_data");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)dx % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_few_bwd_data: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)p.B * p.IH * p.IW;
  const int ppb = few_ppb(npix, p.Cin);
  const dim3 grid((unsigned)cdiv(npix, ppb));
  ProfScope ps(K_FEW_BWD_DATA, ((double)npix * p.Cin + (double)p.B * p.OH * p.OW * 4) * 4, s);
  FEW_DISPATCH(k_few_bwd_data, grid, dim3(256), 0, s, p, dy, w, dx, npix, ppb);
  return check_launch("csg_conv_few_bwd_data");
}

int64_t csg_conv_few_bwd_weight_workspace(const csg_few_desc* d) {
  FewParams p;
  if (few_plan(d, p, "csg_conv_few_bwd_weight_workspace")) return -1;
  const int64_t npix = (int64_t)p.B * p.IH * p.IW;
  const int64_t nblk = cdiv(npix, few_ppb(npix, p.Cin));
  return nblk * ((int64_t)4 * d->KH * d->KW * p.Cin + 4) * 4;
}

int csg_conv_few_bwd_weight(const csg_few_desc* d, const float* x, const float* dy, float* dw, float* db,
                            float* workspace, int64_t workspace_bytes, void* stream) {
  FewParams p;
  int rc = few_plan(d, p, "csg_conv_few_bwd_weight");
  if (rc) return rc;
  const int64_t need = csg_conv_few_bwd_weight_workspace(d);
  CSG_REQUIRE(workspace != nullptr && workspace_bytes >= need, CSG_E_WORKSPACE, "csg_conv_few_bwd_weight: needs %lld bytes",
              (long long)need);
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0 &&
                  ((uintptr_t)workspace % 16) == 0 && (db == nullptr || ((uintptr_t)db % 16) == 0),
              CSG_E_UNSUPPORTED, "csg_conv_few_bwd_weight: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t npix = (int64_t)p.B * p.IH * p.IW, nout = (int64_t)p.B * p.OH * p.OW;
  const int ppb = few_ppb(npix, p.Cin);
  const int64_t nblk = cdiv(npix, ppb);
  const int opb = (int)cdiv(nout, nblk);
  const int64_t wsize = (int64_t)4 * d->KH * d->KW * p.Cin;
  float* dbslabs = db != nullptr ? workspace + nblk * wsize : nullptr;
  const dim3 grid((unsigned)nblk);
  ProfScope ps(K_FEW_BWD_WEIGHT, ((double)npix * p.Cin + (double)nout * 4) * 4, s);
  FEW_DISPATCH(k_few_bwd_weight, grid, dim3(256), 0, s, p, x, dy, workspace, dbslabs, npix, ppb, nout,
               opb);
  rc = check_launch("csg_conv_few_bwd_weight");
  if (rc) return rc;
  launch_slab_reduce(workspace, wsize, dw, dbslabs, db != nullptr ? 4 : 0, db, (int)nblk, s);
  return check_launch("csg_conv_few_bwd_weight(slab sum)");
}

}  // extern "C"
