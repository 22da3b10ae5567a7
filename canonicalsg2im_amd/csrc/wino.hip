// K8w — 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2,3x3) on the fp32 matrix cores of gfx950.
//
// Reference call sites: the 3x3 nn.Conv2d layers of the generator — spade/models/networks/generator.py:28,
// architecture.py:29-31 (conv_0 / conv_1), normalization.py:89-94 (mlp_shared, mlp_gamma, mlp_beta) — and their
// backward-data passes (a 3x3 convolution of dY with the flipped, transposed weights).  70 % of the step's FLOPs
// are these layers; Winograd's minimal filtering computes a 2x2 output tile from a 4x4 input tile with 16
// multiplications per (cin, cout) pair instead of 36: 2.25x fewer MFMA cycles for the same fp32 result
// (error ~2x that of the direct sum, far inside the 1e-4 contract: tests/test_gpu_wino.py).
//
//     U = G g G^T   (weights, 4x4 per (cout,cin), packed once per weight version by k_wino_pack)
//     V = B^T d B   (input tile, formed in registers right before the MFMAs)
//     M_p = sum_cin V_p U_p  for the 16 positions p = (xi, nu)   <- 16 independent GEMMs on v_mfma_f32_32x32x2_f32
//     Y = A^T M A   (2x2 outputs; bias / activation / residual fused behind it)
//
// Work decomposition.  A block owns 64 tiles (TW x TH, 128..256 output pixels of one image) x 64 output channels
// x all 16 positions.  Wave w owns the four positions of row xi = w: 4 nu x 2 tile-groups x 2 channel-groups =
// 16 accumulators of 32x32 (256 VGPRs; one wave per SIMD, one block per CU — the fp32 MFMA issues back to back
// from a single wave, it needs no second wave to hide latency).
//   * B operand (U): every (position, channel-group) is read by exactly ONE wave, so it never goes through LDS:
//     k_wino_pack stores U in the MFMA operand order [xi][nu][cout/32][cin/8][lane][4] and a lane fetches its
//     operands of four consecutive MFMAs with one coalesced 16-byte load (L2 resident: 16*Cin*Cout*4 B per layer).
//   * A operand (V): the raw input region ((2TH+2) x (2TW+2) pixels x 16 channels) is staged in LDS with
//     even/odd columns de-interleaved and 18-word pixel rows, so that the eight ds_read_b64 a lane needs for its
//     tile (two rows x four columns — row xi of B^T d touches two input rows) are bank-conflict free; the
//     transform is 8 VALU adds per 4 MFMA operands.
//   * K loop: 16 channels per LDS stage (128 MFMAs per wave), double-buffered, global loads of stage s+1 issued
//     before the MFMAs of stage s and written to LDS after them: one barrier per 8192 matrix-pipe cycles.
//   * Epilogue: each wave reduces its row over nu (M A), the four rows meet in LDS (A^T .), 16-byte stores.
#include <stddef.h>

#include "csg_common.h"

using namespace csg;

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WN_BK 16      // channels per LDS stage
#define WN_PS 18      // words per staged pixel: 16 channels + 2 (8-byte aligned; 18*l mod 64 visits every even bank once)
#define WN_RSE 36     // words per tile row of the epilogue exchange buffer (32 channels + 4)
#define WN_MAXLD 7    // float4 global loads per thread and stage (396 pixels x 4 / 256)

struct WinoParams {
  int B, H, W, Cin, x_cs, Cout, y_cs;
  int TW, TH;          // tiles per block region (TW * TH == 64)
  int tbx, tby;        // block regions per image
  int nblocks;         // ceil(Cout / 64)
  int RS;              // LDS row stride in words
  int NT32, Q8;        // extents of the packed weights
  int act;
  float slope;
  int nstage;          // ceil(Cin / 16)
};

__device__ __forceinline__ int wn_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------ weight packing
// up[(((xi*4+nu)*NT32 + nt)*Q8 + q)*64 + lane] (float4) = U[xi][nu][n = nt*32 + (lane&31)][k], k = 8q + 2h + {0,1}
// (.x,.y) and 8q + 4 + 2h + {0,1} (.z,.w), h = lane>>5; zero beyond N / K.  The weight is read through element
// strides (s_n, s_k, s_h, s_w) so that the same kernel packs the forward operand (n = cout, k = cin) and the
// backward-data operand (n = cin, k = cout, taps flipped).  `sigma` (nullable) divides every weight first
// (W / sigma of spectral normalisation, rounded as the reference rounds it).
__global__ __launch_bounds__(256) void k_wino_pack(const float* __restrict__ w, int64_t s_n, int64_t s_k, int64_t s_h,
                                                    int64_t s_w, int flip, int N, int K, const float* __restrict__ sigma,
                                                    int NT32, int Q8, float4* __restrict__ up) {
  const int64_t total = (int64_t)16 * NT32 * Q8 * 64;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  int64_t rest = idx >> 6;
  const int q = (int)(rest % Q8);
  rest /= Q8;
  const int nt = (int)(rest % NT32);
  const int p = (int)(rest / NT32);
  const int xi = p >> 2, nu = p & 3;
  const int n = nt * 32 + (lane & 31), h = lane >> 5;
  const float sg = sigma != nullptr ? sigma[0] : 1.0f;
  // rows of G: (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1)
  const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  float out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = 8 * q + (e >> 1) * 4 + 2 * h + (e & 1);
    float u = 0.f;
    if (n < N && k < K) {
      const float* g = w + (int64_t)n * s_n + (int64_t)k * s_k;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        float t = 0.f;                              // t = sum_b g[a][b] * G[nu][b]
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int aa = flip ? 2 - a : a, bb = flip ? 2 - b : b;
          float gv = g[aa * s_h + bb * s_w];
          if (sigma != nullptr) gv = gv / sg;
          t += gv * G[nu][b];
        }
        u += G[xi][a] * t;
      }
    }
    out[e] = u;
  }
  up[idx] = make_float4(out[0], out[1], out[2], out[3]);
}

// ------------------------------------------------------------------------------------ convolution
__global__ __launch_bounds__(256, 1) void k_wino_conv(WinoParams p, const float* __restrict__ x,
                                                       const float4* __restrict__ up, const float* __restrict__ bias,
                                                       const float* __restrict__ res, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int R = 2 * p.TH + 2, C = 2 * p.TW + 2;
  const int bufw = R * p.RS;                     // words per input buffer

  // ---- block -> (image, region, channel block); channel blocks of one region are adjacent (same XCD: input reuse)
  int bid = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % p.nblocks;
  bid /= p.nblocks;
  const int bx = bid % p.tbx;
  bid /= p.tbx;
  const int by = bid % p.tby;
  const int img = bid / p.tby;
  const int X0 = bx * 2 * p.TW, Y0 = by * 2 * p.TH;          // first output pixel of the region

  // ---- staging plan of this thread (k-invariant): global offset (floats, channel 0) and LDS word offset
  int goff[WN_MAXLD], loff[WN_MAXLD];
  const int nld = R * C * 4;
#pragma unroll
  for (int i = 0; i < WN_MAXLD; ++i) {
    const int e = tid + 256 * i;
    goff[i] = -1;
    loff[i] = -1;
    if (e < nld) {
      const int pix = e >> 2, c4 = e & 3;
      const int row = pix / C, col = pix - row * C;
      const int iy = Y0 + row - 1, ix = X0 + col - 1;
      loff[i] = row * p.RS + ((col & 1) * (C >> 1) + (col >> 1)) * WN_PS + c4 * 4;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) goff[i] = ((img * p.H + iy) * p.W + ix) * p.x_cs + c4 * 4;
    }
  }
  float4 st[WN_MAXLD];
  auto load_stage = [&](int s) {
    const int kb = s * WN_BK;
#pragma unroll
    for (int i = 0; i < WN_MAXLD; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      // Cin % 4 == 0: a float4 never straddles the channel end
      if (goff[i] >= 0 && kb + (((tid + 256 * i) & 3) << 2) < p.Cin) v = *(const float4*)(x + (int64_t)goff[i] + kb);
      st[i] = v;
    }
  };
  auto store_stage = [&](int buf) {
    float* base = smem + buf * bufw;
#pragma unroll
    for (int i = 0; i < WN_MAXLD; ++i) {
      if (loff[i] >= 0) {
        *(float2*)(base + loff[i]) = make_float2(st[i].x, st[i].y);
        *(float2*)(base + loff[i] + 2) = make_float2(st[i].z, st[i].w);
      }
    }
  };

  // ---- this lane's tile inside each of the two tile groups, and the two input rows its wave combines
  const int j = lane & 31, h = lane >> 5;
  const int tx = j % p.TW, tyl = j / p.TW;
  const int rows_per_group = 32 / p.TW;
  // row xi of B^T d:  xi=0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
  const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  int aoff[2][2];                                // [group][row a/b]: word offset of column 0 (even half), channel pair h
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int ty = mt * rows_per_group + tyl;
    aoff[mt][0] = (2 * ty + ia) * p.RS + tx * WN_PS + 2 * h;
    aoff[mt][1] = (2 * ty + ib) * p.RS + tx * WN_PS + 2 * h;
  }
  const int half = (C >> 1) * WN_PS;             // odd columns live `half` words after the even ones

  // ---- packed weights of this wave: [xi = wave][nu][nt32][q][lane]
  int uoff[4][2];
  bool uok[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nt32 = nb * 2 + nt;
    uok[nt] = nt32 < p.NT32;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) uoff[nu][nt] = (((wave * 4 + nu) * p.NT32 + (uok[nt] ? nt32 : 0)) * p.Q8) * 64 + lane;
  }
  float4 ua[4][2], ub[4][2];
  auto load_u = [&](float4 (&u)[4][2], int q) {
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        u[nu][nt] = (uok[nt] && q < p.Q8) ? up[uoff[nu][nt] + q * 64] : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  f32x16 acc[4][2][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nu][mt][nt][e] = 0.f;

  // one k-oct (8 channels) of one stage: 16 ds_read_b64 + 16 VALU per tile group, then 32 MFMAs per group
  auto compute_oct = [&](const float* buf, int o, const float4 (&u)[4][2]) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float2 v[4][2];                            // [nu][channel pair]
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const float* pa = buf + aoff[mt][0] + 8 * o + 4 * cp;
        const float* pb = buf + aoff[mt][1] + 8 * o + 4 * cp;
        float2 r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {            // column c of the 4x4 patch: even/odd halves, then + c>>1 pixels
          const int co = (c & 1) * half + (c >> 1) * WN_PS;
          const float2 da = *(const float2*)(pa + co), db = *(const float2*)(pb + co);
          r[c] = make_float2(da.x + sgn * db.x, da.y + sgn * db.y);
        }
        v[0][cp] = make_float2(r[0].x - r[2].x, r[0].y - r[2].y);
        v[1][cp] = make_float2(r[1].x + r[2].x, r[1].y + r[2].y);
        v[2][cp] = make_float2(r[2].x - r[1].x, r[2].y - r[1].y);
        v[3][cp] = make_float2(r[1].x - r[3].x, r[1].y - r[3].y);
      }
#pragma unroll
      for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          // weights as the first operand: D[i = channel][j = tile] -> a lane holds one tile and runs of 4 channels
          acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].x, v[nu][0].x, acc[nu][mt][nt], 0, 0, 0);
          acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].y, v[nu][0].y, acc[nu][mt][nt], 0, 0, 0);
          acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].z, v[nu][1].x, acc[nu][mt][nt], 0, 0, 0);
          acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].w, v[nu][1].y, acc[nu][mt][nt], 0, 0, 0);
        }
    }
  };

  // ---- K loop
  load_stage(0);
  load_u(ua, 0);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < p.nstage; ++s) {
    const float* buf = smem + (s & 1) * bufw;
    const bool more = s + 1 < p.nstage;
    if (more) load_stage(s + 1);
    load_u(ub, 2 * s + 1);
    compute_oct(buf, 0, ua);
    if (more) load_u(ua, 2 * s + 2);
    compute_oct(buf, 1, ub);
    if (more) store_stage((s + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: M A per wave (row xi), A^T . across the four waves through LDS, one channel group at a time
  //   R[0] = M0 + M1 + M2,  R[1] = M1 - M2 - M3;   Y[0][b] = R0b + R1b + R2b,  Y[1][b] = R1b - R2b - R3b
  float* rbuf = smem;                            // [xi][b][64 tiles][WN_RSE]
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float r0[4], r1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float m0 = acc[0][mt][nt][4 * g + e], m1 = acc[1][mt][nt][4 * g + e], m2 = acc[2][mt][nt][4 * g + e],
                      m3 = acc[3][mt][nt][4 * g + e];
          r0[e] = m0 + m1 + m2;
          r1[e] = m1 - m2 - m3;
        }
        const int tile = mt * 32 + j, ch = 8 * g + 4 * h;
        *(float4*)(rbuf + ((wave * 2 + 0) * 64 + tile) * WN_RSE + ch) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        *(float4*)(rbuf + ((wave * 2 + 1) * 64 + tile) * WN_RSE + ch) = make_float4(r1[0], r1[1], r1[2], r1[3]);
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 256 * it;           // 64 tiles x 8 channel quads
      const int tile = item >> 3, cq = item & 7;
      const int n = nb * 64 + nt * 32 + cq * 4;
      const int mt = tile >> 5, jj = tile & 31;
      const int ttx = jj % p.TW, tty = mt * rows_per_group + jj / p.TW;
      const int oy = Y0 + 2 * tty, ox = X0 + 2 * ttx;
      if (n < p.Cout && oy < p.H && ox < p.W) {
        float4 rr[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
          for (int b = 0; b < 2; ++b) rr[xi][b] = *(const float4*)(rbuf + ((xi * 2 + b) * 64 + tile) * WN_RSE + cq * 4);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) bv = *(const float4*)(bias + n);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float v[4];
            const float4 q0 = rr[0][b], q1 = rr[1][b], q2 = rr[2][b], q3 = rr[3][b];
            if (a == 0) {
              v[0] = q0.x + q1.x + q2.x; v[1] = q0.y + q1.y + q2.y; v[2] = q0.z + q1.z + q2.z; v[3] = q0.w + q1.w + q2.w;
            } else {
              v[0] = q1.x - q2.x - q3.x; v[1] = q1.y - q2.y - q3.y; v[2] = q1.z - q2.z - q3.z; v[3] = q1.w - q2.w - q3.w;
            }
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (p.act == CSG_ACT_LEAKY)
                v[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
              else if (p.act == CSG_ACT_TANH)
                v[e] = tanhf(v[e]);
            }
            // H and W are even: a tile is either wholly inside the image or wholly outside
            const int64_t pix = ((int64_t)img * p.H + (oy + a)) * p.W + (ox + b);
            if (res != nullptr) {
              const float4 rv = *(const float4*)(res + pix * p.y_cs + n);
              v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            }
            *(float4*)(y + pix * p.y_cs + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ host side
static int wn_row_stride(int TW) {
  // smallest row stride >= C*PS (even) for which the 32 lanes of a ds_read_b64 group — tile (tx, tyl), word offset
  // PS*tx + 2*RS*tyl — fall on 32 different bank pairs (bank = word address mod 64)
  const int C = 2 * TW + 2;
  for (int pad = 0; pad < 256; pad += 2) {
    const int RS = C * WN_PS + pad;
    unsigned long long used = 0;
    bool ok = true;
    for (int j = 0; j < 32 && ok; ++j) {
      const int wa = (WN_PS * (j % TW) + 2 * RS * (j / TW)) & 63;
      const unsigned long long m = (1ull << wa) | (1ull << ((wa + 1) & 63));
      if (used & m) ok = false;
      used |= m;
    }
    if (ok) return RS;
  }
  return C * WN_PS;
}

static int wn_plan(const csg_wino_desc* d, WinoParams& p, size_t& shm, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  CSG_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && d->W >= 8, CSG_E_UNSUPPORTED, "%s: H=%d, W=%d must be even, W >= 8", who,
              d->H, d->W);
  CSG_REQUIRE(d->Cin % 4 == 0 && d->x_cs % 4 == 0 && d->x_cs >= d->Cin && d->Cout % 4 == 0 && d->y_cs % 4 == 0 &&
                  d->y_cs >= d->Cout,
              CSG_E_UNSUPPORTED, "%s: channel counts and strides must be multiples of 4", who);
  CSG_REQUIRE((int64_t)d->B * d->H * d->W * (int64_t)(d->x_cs > d->y_cs ? d->x_cs : d->y_cs) < (1ll << 31), CSG_E_UNSUPPORTED,
              "%s: tensor too large for 32-bit offsets", who);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.Cout = d->Cout; p.y_cs = d->y_cs;
  p.TW = d->W >= 64 ? 32 : (d->W >= 32 ? 16 : (d->W >= 16 ? 8 : 4));
  p.TH = 64 / p.TW;
  p.tbx = (d->W / 2 + p.TW - 1) / p.TW;
  p.tby = (d->H / 2 + p.TH - 1) / p.TH;
  p.nblocks = (d->Cout + 63) / 64;
  p.RS = wn_row_stride(p.TW);
  p.NT32 = (d->Cout + 31) / 32;
  p.Q8 = (d->Cin + 7) / 8;
  p.act = d->act; p.slope = d->slope;
  p.nstage = (d->Cin + WN_BK - 1) / WN_BK;
  const size_t in_bytes = (size_t)2 * (2 * p.TH + 2) * p.RS * 4;
  const size_t ep_bytes = (size_t)4 * 2 * 64 * WN_RSE * 4;
  shm = in_bytes > ep_bytes ? in_bytes : ep_bytes;
  CSG_REQUIRE((2 * p.TH + 2) * (2 * p.TW + 2) * 4 <= WN_MAXLD * 256, CSG_E_UNSUPPORTED, "%s: staging plan too large", who);
  return CSG_OK;
}

extern "C" {

int64_t csg_wino_pack_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return -1;
  return (int64_t)16 * cdiv(N, 32) * cdiv(K, 8) * 64 * 16;
}

int csg_wino_pack_weights(const float* w, int64_t Cout, int64_t Cin, int32_t backward_data, const float* sigma,
                          float* packed, void* stream) {
  CSG_REQUIRE(w != nullptr && packed != nullptr && Cout > 0 && Cin > 0, CSG_E_BADSHAPE, "csg_wino_pack_weights: bad arguments");
  CSG_REQUIRE(((uintptr_t)packed % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino_pack_weights: packed must be 16-byte aligned");
  // w is (Cout, Cin, 3, 3) contiguous.  forward: n = cout, k = cin;  backward-data: n = cin, k = cout, taps flipped
  const int64_t N = backward_data ? Cin : Cout, K = backward_data ? Cout : Cin;
  const int64_t s_n = backward_data ? 9 : Cin * 9, s_k = backward_data ? Cin * 9 : 9;
  const int NT32 = (int)cdiv(N, 32), Q8 = (int)cdiv(K, 8);
  const int64_t total = (int64_t)16 * NT32 * Q8 * 64;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(K_WINO_PACK, (double)Cout * Cin * 9 * 4 + (double)total * 16, s);
  hipLaunchKernelGGL(k_wino_pack, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, w, s_n, s_k, (int64_t)3, (int64_t)1,
                     backward_data ? 1 : 0, (int)N, (int)K, sigma, NT32, Q8, (float4*)packed);
  return check_launch("csg_wino_pack_weights");
}

int csg_wino_conv(const csg_wino_desc* d, const float* x, const float* packed, const float* bias, const float* residual,
                  float* y, void* stream) {
  WinoParams p;
  size_t shm = 0;
  int rc = wn_plan(d, p, shm, "csg_wino_conv");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_wino_conv: pointers must be 16-byte aligned");
  static bool attr_set[16] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wino_conv, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_wino_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    attr_set[dev] = true;
  }
  CSG_REQUIRE(shm <= 96 * 1024, CSG_E_UNSUPPORTED, "csg_wino_conv: %zu bytes of LDS", shm);
  hipStream_t s = (hipStream_t)stream;
  const int64_t grid = (int64_t)p.B * p.tby * p.tbx * p.nblocks;
  CSG_REQUIRE(grid < (1ll << 31), CSG_E_UNSUPPORTED, "csg_wino_conv: grid too large");
  // algorithmic FLOPs of the DIRECT convolution this replaces (2 * M * 9*Cin * Cout): what FlopCounterMode counts
  ProfScope ps(K_WINO_CONV, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, s);
  hipLaunchKernelGGL(k_wino_conv, dim3((unsigned)grid), dim3(256), shm, s, p, x, (const float4*)packed, bias, residual, y);
  return check_launch("csg_wino_conv");
}

}  // extern "C"
