// K3/K8/K11 — implicit-GEMM convolution on the fp32 matrix cores of gfx950.
//
// Reference call sites: nn.Conv2d in spade/models/networks/generator.py:28,46,
// architecture.py:29-32, normalization.py:89-94, discriminator.py:175-187; nn.Linear in
// sg2im/layers.py:10.  The reference computes in fp32, and v_mfma_f32_32x32x2_f32 is an exact
// k-ordered fp32 FMA chain, so this path keeps the reference's precision (no TF32/bf16).
//
// GEMM view (forward and backward-data): rows m = output pixels, cols n = output channels,
// k = (tap, input channel).  Block tile 128 x BN x 32, 4 waves (2x2), each wave 64 x BN/2 as
// 2 x NI tiles of 32x32.  Both operands are staged global -> registers -> LDS as [row][32 k]
// with rows padded to 36 floats: every fragment read is one conflict-free ds_read_b128 that
// feeds FOUR MFMAs (the k index inside a group of 8 is permuted identically for A and B).
// LDS is double-buffered: one barrier per K-tile, next tile's global loads in flight during
// the 64 MFMAs of the current one; 2 blocks/CU so a second wave covers each SIMD's gaps.
//
// Weight gradient: rows = output channels (A = dY), cols = (tap, input channel) (B = gathered
// X), reduction over pixels, split across blocks into slabs that an ordered pass sums
// (bit-reproducible, no atomics).
#include "csg_common.h"

using namespace csg;

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct IgemmParams {
  csg_conv_desc d;
  int M, Ktot, wrow;
  int mtiles, ntiles;
  int simple_out;  // output pixel index == m
};

#define IG_BM 128
#define IG_BK 32
#define IG_LD 36

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ void load_taps(const csg_conv_desc& d, int* s_tap, int tid) {
#pragma unroll
  for (int i = 0; i < CSG_MAX_TAPS; ++i) {
    if (tid == i) {
      s_tap[i] = d.tap_dy[i];
      s_tap[16 + i] = d.tap_dx[i];
      s_tap[32 + i] = d.tap_w[i];
    }
  }
}

template <int BN>
__global__ __launch_bounds__(256, 2) void k_igemm_fwd(IgemmParams p, const float* __restrict__ x,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ y) {
  constexpr int NI = BN / 64;
  constexpr int BROWS = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * IG_BM * IG_LD;
  int* s_tap = (int*)(Bs + 2 * BN * IG_LD);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;
  const csg_conv_desc& d = p.d;

  load_taps(d, s_tap, tid);

  const int r0 = tid >> 3, kc = tid & 7;
  int a_rb[4], a_iy0[4], a_ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = mt * IG_BM + r0 + 32 * i;
    bool ok = m < p.M;
    int mm = ok ? m : 0;
    int t = mm / d.OWg;
    int gx = mm - t * d.OWg;
    int b = t / d.OHg;
    int gy = t - b * d.OHg;
    a_rb[i] = b * d.IHp;
    a_iy0[i] = ok ? gy * d.istride : -(1 << 28);
    a_ix0[i] = gx * d.istride;
  }
  int b_n[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) b_n[i] = nt * BN + r0 + 32 * i;

  float4 ra[4], rb[BROWS];
  const int nkt = (p.Ktot + IG_BK - 1) / IG_BK;
  __syncthreads();  // s_tap visible

  auto load_tile = [&](int kt) {
    const int k0 = kt * IG_BK + kc * 4;
    const bool kv = k0 < p.Ktot;
    const int slot = kv ? k0 / d.Cin : 0;
    const int c = k0 - slot * d.Cin;
    const int dy = s_tap[slot], dx = s_tap[16 + slot], tw = s_tap[32 + slot];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      const bool inb = kv && (unsigned)iy < (unsigned)d.IHv && (unsigned)ix < (unsigned)d.IWv;
      const int64_t off = ((int64_t)(a_rb[i] + (iy >> d.in_up)) * d.IWp + (ix >> d.in_up)) * d.x_cs + c;
      ra[i] = inb ? *(const float4*)(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t wcol = (int64_t)tw * d.Cin + c;
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const bool nb = kv && b_n[i] < d.Cout;
      rb[i] = nb ? *(const float4*)(w + (int64_t)b_n[i] * p.wrow + wcol) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
    float* a = As + buf * IG_BM * IG_LD + r0 * IG_LD + kc * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(float4*)(a + 32 * i * IG_LD) = ra[i];
    float* b = Bs + buf * BN * IG_LD + r0 * IG_LD + kc * 4;
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *(float4*)(b + 32 * i * IG_LD) = rb[i];
  };

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, hh = lane >> 5;

  load_tile(0);
  store_tile(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_tile(kt + 1);
    const float* Ab = As + buf * IG_BM * IG_LD + (wm * 64 + r) * IG_LD + 4 * hh;
    const float* Bb = Bs + buf * BN * IG_LD + (wn * (BN / 2) + r) * IG_LD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a[2], b[NI];
      a[0] = *(const float4*)(Ab + g * 8);
      a[1] = *(const float4*)(Ab + 32 * IG_LD + g * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const float4*)(Bb + ni * 32 * IG_LD + g * 8);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].x, b[ni].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].y, b[ni].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].z, b[ni].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi].w, b[ni].w, acc[mi][ni], 0, 0, 0);
        }
    }
    if (kt + 1 < nkt) store_tile(buf ^ 1);
    __syncthreads();
  }

  // epilogue: D[i][j], j = lane&31 (output channel), i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (pixel)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = nt * BN + wn * (BN / 2) + ni * 32 + r;
    const bool nok = n < d.Cout;
    const float bv = (bias != nullptr && nok) ? bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = mt * IG_BM + wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (nok && m < p.M) {
          int64_t pix = m;
          if (!p.simple_out) {
            int t = m / d.OWg;
            int gx = m - t * d.OWg;
            int b = t / d.OHg;
            int gy = t - b * d.OHg;
            pix = ((int64_t)b * d.OHf + gy * d.os + d.ooy) * d.OWf + gx * d.os + d.oox;
          }
          const int64_t off = pix * d.y_cs + n;
          float v = acc[mi][ni][e] + bv;
          if (d.act == CSG_ACT_LEAKY)
            v = v > 0.f ? v : v * d.slope;
          else if (d.act == CSG_ACT_TANH)
            v = tanhf(v);
          if (res != nullptr) v += res[off];
          if (d.accumulate) v += y[off];
          y[off] = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------- weight grad
// tile: 128 output channels (i) x 128 (tap,cin) columns (j), 32 pixels per reduction step
#define WG_LD 128
__global__ __launch_bounds__(256, 2) void k_igemm_wgrad(IgemmParams p, const float* __restrict__ x,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         int itiles, int jtiles, int nsplit, int cps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][128]  dY tile, [pixel][n]
  float* Bs = smem + 2 * 32 * WG_LD;     // [2][32][128]  X tile,  [pixel][kk]
  int* s_tap = (int*)(Bs + 2 * 32 * WG_LD);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const csg_conv_desc& d = p.d;
  int bid = blockIdx.x;
  const int sp = bid % nsplit;
  bid /= nsplit;
  const int jt = bid % jtiles, it = bid / jtiles;

  load_taps(d, s_tap, tid);
  __syncthreads();

  const int pr = tid >> 5, c4 = tid & 31;
  // this thread's column of the X tile: fixed (tap, channel)
  const int kk0 = jt * 128 + c4 * 4;
  const bool kv = kk0 < p.Ktot;
  const int slot = kv ? kk0 / d.Cin : 0;
  const int cch = kk0 - slot * d.Cin;
  const int tdy = s_tap[slot], tdx = s_tap[16 + slot];
  // this thread's column of the dY tile
  const int n0 = it * 128 + c4 * 4;
  const bool nv = n0 < d.Cout;  // Cout % 4 == 0

  const int nch = (p.M + 31) / 32;
  const int ch0 = sp * cps, ch1 = min(nch, ch0 + cps);

  float4 ra[4], rb[4];
  auto load_tile = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = ch * 32 + pr + 8 * i;
      const bool ok = m < p.M;
      ra[i] = (ok && nv) ? *(const float4*)(dy + (int64_t)m * d.y_cs + n0) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int mm = ok ? m : 0;
      const int t = mm / d.OWg;
      const int gx = mm - t * d.OWg;
      const int b = t / d.OHg;
      const int gy = t - b * d.OHg;
      const int iy = gy * d.istride + tdy, ix = gx * d.istride + tdx;
      const bool inb = ok && kv && (unsigned)iy < (unsigned)d.IHv && (unsigned)ix < (unsigned)d.IWv;
      const int64_t off = ((int64_t)(b * d.IHp + (iy >> d.in_up)) * d.IWp + (ix >> d.in_up)) * d.x_cs + cch;
      rb[i] = inb ? *(const float4*)(x + off) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(float4*)(As + buf * 32 * WG_LD + (pr + 8 * i) * WG_LD + c4 * 4) = ra[i];
      *(float4*)(Bs + buf * 32 * WG_LD + (pr + 8 * i) * WG_LD + c4 * 4) = rb[i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int wi = wave >> 1, wj = wave & 1, r = lane & 31, hh = lane >> 5;

  if (ch0 < ch1) {
    load_tile(ch0);
    store_tile(0);
  }
  __syncthreads();
  for (int ch = ch0; ch < ch1; ++ch) {
    const int buf = (ch - ch0) & 1;
    if (ch + 1 < ch1) load_tile(ch + 1);
    const float* Ab = As + buf * 32 * WG_LD + hh * WG_LD + wi * 64 + r;
    const float* Bb = Bs + buf * 32 * WG_LD + hh * WG_LD + wj * 64 + r;
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      const float a0 = Ab[kp * 2 * WG_LD], a1 = Ab[kp * 2 * WG_LD + 32];
      const float b0 = Bb[kp * 2 * WG_LD], b1 = Bb[kp * 2 * WG_LD + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (ch + 1 < ch1) store_tile(buf ^ 1);
    __syncthreads();
  }

  // D[i = n][j = kk]; slab layout [split][Cout][wrow] with the weight-tap index applied
  float* slab = out + (int64_t)sp * d.Cout * p.wrow;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int kk = jt * 128 + wj * 64 + ni * 32 + r;
    if (kk < p.Ktot) {
      const int sl = kk / d.Cin;
      const int col = s_tap[32 + sl] * d.Cin + (kk - sl * d.Cin);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = it * 128 + wi * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          if (n < d.Cout) slab[(int64_t)n * p.wrow + col] = acc[mi][ni][e];
        }
      }
    }
  }
}

__global__ void k_wgrad_reduce(const float* __restrict__ ws, int64_t n4, int nsplit, float* __restrict__ dw) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 a = ((const float4*)ws)[i];
  for (int s = 1; s < nsplit; ++s) {
    float4 b = ((const float4*)ws)[(int64_t)s * n4 + i];
    a.x += b.x;
    a.y += b.y;
    a.z += b.z;
    a.w += b.w;
  }
  ((float4*)dw)[i] = a;
}

// ------------------------------------------------------------------------------- host side
static int validate(const csg_conv_desc* d, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->IHp > 0 && d->IWp > 0 && d->Cin > 0 && d->Cout > 0 && d->OHg > 0 && d->OWg > 0,
              CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  CSG_REQUIRE(d->Cin % 4 == 0 && d->x_cs % 4 == 0 && d->x_cs >= d->Cin, CSG_E_UNSUPPORTED,
              "%s: Cin=%d and x_cs=%d must be multiples of 4 (16-byte channel rows)", who, d->Cin, d->x_cs);
  CSG_REQUIRE(d->ntaps >= 1 && d->ntaps <= CSG_MAX_TAPS && d->wtaps >= 1, CSG_E_UNSUPPORTED, "%s: ntaps=%d", who,
              d->ntaps);
  CSG_REQUIRE(d->in_up >= 0 && d->in_up <= 4 && d->os >= 1 && d->istride >= 1, CSG_E_BADSHAPE, "%s: bad strides", who);
  CSG_REQUIRE(d->y_cs >= d->Cout, CSG_E_BADSHAPE, "%s: y_cs < Cout", who);
  for (int t = 0; t < d->ntaps; ++t)
    CSG_REQUIRE(d->tap_w[t] >= 0 && d->tap_w[t] < d->wtaps, CSG_E_BADSHAPE, "%s: tap_w out of range", who);
  int64_t M = (int64_t)d->B * d->OHg * d->OWg;
  CSG_REQUIRE(M < (1ll << 31) - 256, CSG_E_UNSUPPORTED, "%s: too many output pixels", who);
  CSG_REQUIRE((int64_t)d->B * d->OHf * d->OWf < (1ll << 31), CSG_E_UNSUPPORTED, "%s: output too large", who);
  return CSG_OK;
}

static void fill(IgemmParams& p, const csg_conv_desc* d) {
  p.d = *d;
  p.M = d->B * d->OHg * d->OWg;
  p.Ktot = d->ntaps * d->Cin;
  p.wrow = d->wtaps * d->Cin;
  p.simple_out = (d->os == 1 && d->ooy == 0 && d->oox == 0 && d->OHg == d->OHf && d->OWg == d->OWf) ? 1 : 0;
}

template <int BN>
static int launch_fwd(IgemmParams& p, const float* x, const float* w, const float* bias, const float* res, float* y,
                      hipStream_t s) {
  p.mtiles = (p.M + IG_BM - 1) / IG_BM;
  p.ntiles = (p.d.Cout + BN - 1) / BN;
  static bool attr_set = false;
  size_t shm = (size_t)(2 * IG_BM * IG_LD + 2 * BN * IG_LD) * 4 + 48 * 4;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_igemm_fwd<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  hipLaunchKernelGGL(k_igemm_fwd<BN>, dim3((unsigned)(p.mtiles * p.ntiles)), dim3(256), shm, s, p, x, w, bias, res, y);
  return check_launch("csg_conv_fwd");
}

static void wgrad_plan(const IgemmParams& p, int& itiles, int& jtiles, int& nsplit, int& cps) {
  itiles = (p.d.Cout + 127) / 128;
  jtiles = (p.Ktot + 127) / 128;
  const int nch = (p.M + 31) / 32;
  int base = itiles * jtiles;
  nsplit = 1024 / base;
  if (nsplit < 1) nsplit = 1;
  int maxsplit = nch / 8;
  if (maxsplit < 1) maxsplit = 1;
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit > 256) nsplit = 256;
  cps = (nch + nsplit - 1) / nsplit;
  nsplit = (nch + cps - 1) / cps;  // no empty splits
}

extern "C" {

int csg_conv_fwd(const csg_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                 float* y, void* stream) {
  int rc = validate(d, "csg_conv_fwd");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_fwd: x and w must be 16-byte aligned");
  IgemmParams p;
  fill(p, d);
  hipStream_t s = (hipStream_t)stream;
  // algorithmic FLOPs: 2 * M * Ktot * Cout (zero-padding taps included, as FlopCounterMode counts them)
  ProfScope ps(K_IGEMM_FWD, 2.0 * p.M * (double)p.Ktot * d->Cout, s);
  if (d->Cout <= 64) return launch_fwd<64>(p, x, w, bias, residual, y, s);
  return launch_fwd<128>(p, x, w, bias, residual, y, s);
}

int64_t csg_conv_bwd_weight_workspace(const csg_conv_desc* d) {
  if (validate(d, "csg_conv_bwd_weight_workspace")) return -1;
  IgemmParams p;
  fill(p, d);
  int it, jt, ns, cps;
  wgrad_plan(p, it, jt, ns, cps);
  return ns > 1 ? (int64_t)ns * d->Cout * p.wrow * 4 : 0;
}

int csg_conv_bwd_weight(const csg_conv_desc* d, const float* x, const float* dy, float* dw, float* workspace,
                        int64_t workspace_bytes, void* stream) {
  int rc = validate(d, "csg_conv_bwd_weight");
  if (rc) return rc;
  CSG_REQUIRE(d->Cout % 4 == 0 && d->y_cs % 4 == 0, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: Cout=%d and y_cs=%d must be multiples of 4", d->Cout, d->y_cs);
  CSG_REQUIRE(d->os == 1 && d->ooy == 0 && d->oox == 0 && d->OHg == d->OHf && d->OWg == d->OWf, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: needs the forward descriptor (dense output grid)");
  CSG_REQUIRE(d->ntaps == d->wtaps, CSG_E_UNSUPPORTED, "csg_conv_bwd_weight: every weight tap must be listed");
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: pointers must be 16-byte aligned");
  IgemmParams p;
  fill(p, d);
  p.mtiles = p.ntiles = 0;
  int itiles, jtiles, nsplit, cps;
  wgrad_plan(p, itiles, jtiles, nsplit, cps);
  const int64_t need = nsplit > 1 ? (int64_t)nsplit * d->Cout * p.wrow * 4 : 0;
  CSG_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr), CSG_E_WORKSPACE,
              "csg_conv_bwd_weight: workspace %ld < %ld bytes", (long)workspace_bytes, (long)need);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set = false;
  size_t shm = (size_t)(4 * 32 * WG_LD) * 4 + 48 * 4;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)k_igemm_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    attr_set = true;
  }
  {
    ProfScope ps(K_IGEMM_WGRAD, 2.0 * p.M * (double)p.Ktot * d->Cout, s);
    float* out = nsplit > 1 ? workspace : dw;
    hipLaunchKernelGGL(k_igemm_wgrad, dim3((unsigned)(itiles * jtiles * nsplit)), dim3(256), shm, s, p, x, dy, out,
                       itiles, jtiles, nsplit, cps);
    rc = check_launch("csg_conv_bwd_weight");
    if (rc) return rc;
  }
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)d->Cout * p.wrow / 4;
    ProfScope ps(K_WGRAD_REDUCE, (double)(nsplit + 1) * n4 * 16, s);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, s, workspace, n4, nsplit, dw);
    rc = check_launch("csg_conv_bwd_weight(reduce)");
  }
  return rc;
}

}  // extern "C"
