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
#include <stdlib.h>

#include <type_traits>

#include "csg_buffer.h"
#include "csg_common.h"
#include "csg_pack.h"
#include "csg_reduce.h"

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
  float gate_slope;    // epilogue gate: y *= (gate > 0 ? 1 : gate_slope)
  int nstage;          // ceil(Cin / 16)
  int variant;         // 1: 64 tiles per block (k_wino_conv), 2: 32 tiles per block, two blocks per CU (k_wino_conv2)
  int ntb;             // k_wino_conv2: 32-channel groups per block (2, or 1 when Cout <= 32)
  int ksplit, sps;     // k_wino_conv2: input-channel stages cut into ksplit ranges of sps stages, one output slab each
  long long slab;      // floats per slab (B*H*W*y_cs)
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
// One block packs 32 n x 32 k (all 9 taps, all 16 positions): the taps are staged in LDS with the loads running
// along whichever of n / k is the contiguous axis of the parameter (each weight is read from memory once), a thread
// then forms G g G^T for its (n, 4 k) and writes its 16 float4 — 1 KB contiguous per wave and position.
#define WP_LD 33
__device__ __forceinline__ void wn_pack_tile(const float* __restrict__ w, int64_t s_n, int64_t s_k, int64_t s_h,
                                             int64_t s_w, int flip, int N, int K, const float* __restrict__ sigma,
                                             int NT32, int Q8, float4* __restrict__ up, int qb, int nt) {
  __shared__ float g[9][32][WP_LD];
  const int tid = threadIdx.x;                         // k range [32 qb, 32 qb + 32), n range [32 nt, 32 nt + 32)
  const bool k_fast = s_k <= s_n;
  const float sg = sigma != nullptr ? sigma[0] : 1.0f;
  // all 36 loads of a thread are issued before the first LDS store (one load per loop trip costs a full memory
  // latency per trip: ~20 us even for the smallest weight)
  float stage[36];
#pragma unroll
  for (int it = 0; it < 36; ++it) {
    const int t = it >> 2, r = tid + 256 * (it & 3);
    const int nl = k_fast ? (r >> 5) : (r & 31), kl = k_fast ? (r & 31) : (r >> 5);
    const int n = nt * 32 + nl, k = qb * 32 + kl;
    const int a = t / 3, b = t - 3 * a;
    const int aa = flip ? 2 - a : a, bb = flip ? 2 - b : b;
    stage[it] = (n < N && k < K) ? w[(int64_t)n * s_n + (int64_t)k * s_k + aa * s_h + bb * s_w] : 0.f;
  }
#pragma unroll
  for (int it = 0; it < 36; ++it) {
    const int t = it >> 2, r = tid + 256 * (it & 3);
    const int nl = k_fast ? (r >> 5) : (r & 31), kl = k_fast ? (r & 31) : (r >> 5);
    g[t][nl][kl] = sigma != nullptr ? stage[it] / sg : stage[it];
  }
  __syncthreads();
  const int lane = tid & 63, ql = tid >> 6;            // 4 q per block, one per wave
  const int q = qb * 4 + ql;
  if (q >= Q8) return;
  const int nl = lane & 31, h = lane >> 5;
  float u[16][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int kl = 8 * ql + (e >> 1) * 4 + 2 * h + (e & 1);
    float t[4][3];                                     // t = G g  (rows of G: (1,0,0), (1/2,1/2,1/2), (1/2,-1/2,1/2), (0,0,1))
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float g0 = g[b][nl][kl], g1 = g[3 + b][nl][kl], g2 = g[6 + b][nl][kl];
      t[0][b] = g0;
      t[1][b] = 0.5f * g0 + 0.5f * g1 + 0.5f * g2;
      t[2][b] = 0.5f * g0 - 0.5f * g1 + 0.5f * g2;
      t[3][b] = g2;
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      u[xi * 4 + 0][e] = t[xi][0];
      u[xi * 4 + 1][e] = 0.5f * t[xi][0] + 0.5f * t[xi][1] + 0.5f * t[xi][2];
      u[xi * 4 + 2][e] = 0.5f * t[xi][0] - 0.5f * t[xi][1] + 0.5f * t[xi][2];
      u[xi * 4 + 3][e] = t[xi][2];
    }
  }
#pragma unroll
  for (int p = 0; p < 16; ++p)
    up[(((int64_t)p * NT32 + nt) * Q8 + q) * 64 + lane] = make_float4(u[p][0], u[p][1], u[p][2], u[p][3]);
}

__global__ __launch_bounds__(256) void k_wino_pack(const float* __restrict__ w, int64_t s_n, int64_t s_k, int64_t s_h,
                                                    int64_t s_w, int flip, int N, int K, const float* __restrict__ sigma,
                                                    int NT32, int Q8, float4* __restrict__ up) {
  wn_pack_tile(w, s_n, s_k, s_h, s_w, flip, N, K, sigma, NT32, Q8, up, blockIdx.x, blockIdx.y);
}

// several weights per launch: block -> item by the table's block offsets (csg_pack.h)
__global__ __launch_bounds__(256) void k_wino_pack_multi(PackMulti pm) {
  int i = 0;
  while (i + 1 < pm.n && (int)blockIdx.x >= pm.it[i + 1].start) ++i;
  const PackMultiItem& d = pm.it[i];
  const int local = blockIdx.x - d.start, nqb = (d.Q8 + 3) >> 2;
  wn_pack_tile(d.w, d.s_n, d.s_k, d.s_h, d.s_w, d.flip, d.N, d.K, nullptr, d.NT32, d.Q8, d.up, local % nqb, local / nqb);
}

// ------------------------------------------------------------------------------------ convolution
// smallest row stride >= C*PS (even) for which the 32 lanes of a ds_read_b64 group — tile (tx, tyl), word offset
// PS*tx + 2*RS*tyl — fall on 32 different bank pairs (bank = word address mod 64)
constexpr int wn_row_stride(int TW) {
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

// Every VALU instruction of the single wave per SIMD takes issue time away from the fp32 MFMAs (measured: 57 % matrix
// pipe busy with 2 VALU per MFMA in the loop, profiles/archive/r02_pmc_wino.md), so the loop body carries none that is not
// arithmetic of the transform: the tile geometry is a template parameter (LDS offsets become instruction immediates),
// the per-stage advance of the global loads rides in the SCALAR offset of the buffer instructions, and the stage loop
// is unrolled by two so that the LDS buffer is a compile-time constant.
template <int TW>
__global__ __launch_bounds__(256, 1) void k_wino_conv(WinoParams p, const float* __restrict__ x,
                                                       const float4* __restrict__ up, const float* __restrict__ bias,
                                                       const float* __restrict__ res, const float* __restrict__ gate,
                                                       float* __restrict__ y) {
  constexpr int TH = 64 / TW;
  constexpr int R = 2 * TH + 2, C = 2 * TW + 2;
  constexpr int RS = wn_row_stride(TW);
  constexpr int BUFW = R * RS + 16;              // words per input buffer (+ a 16-word dump slot)
  constexpr int HALF = (C / 2) * WN_PS;          // odd columns live HALF words after the even ones
  constexpr int NLD = (R * C * 4 + 255) / 256;   // float4 global loads per thread and stage
  constexpr int RPG = 32 / TW;                   // tile rows per tile group
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- block -> (image, region, channel block); channel blocks of one region are adjacent (same XCD: input reuse)
  int bid = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % p.nblocks;
  bid /= p.nblocks;
  const int bx = bid % p.tbx;
  bid /= p.tbx;
  const int by = bid % p.tby;
  const int img = bid / p.tby;
  const int X0 = bx * 2 * TW, Y0 = by * 2 * TH;              // first output pixel of the region

  // ---- staging plan of this thread (k-invariant): byte offset of channel 0 in x (CSG_OOB_OFF: zero padding or no
  // work) and LDS word offset (threads beyond the region write a dump slot at the end of the buffer).  All loads go
  // through buffer descriptors: no branch, and the channel advance of a stage is the instruction's scalar offset.
  // Channels past Cin inside the last stage are NOT masked: they hold finite activations (or zeros past the end of
  // the tensor) and meet zero weights in the packed operand.
  const csg_i32x4 rsX = csg_make_srd(x, (long long)p.B * p.H * p.W * p.x_cs * 4);
  const csg_i32x4 rsU = csg_make_srd(up, (long long)16 * p.NT32 * p.Q8 * 64 * 16);
  unsigned goff[NLD];
  int loff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i;
    goff[i] = CSG_OOB_OFF;
    loff[i] = BUFW - 16 + (tid & 3) * 4;
    if (e < R * C * 4) {
      const int pix = e >> 2, c4 = e & 3;
      const int row = pix / C, col = pix - row * C;
      const int iy = Y0 + row - 1, ix = X0 + col - 1;
      loff[i] = row * RS + ((col & 1) * (C >> 1) + (col >> 1)) * WN_PS + c4 * 4;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        goff[i] = (unsigned)(((img * p.H + iy) * p.W + ix) * p.x_cs + c4 * 4) * 4u;
    }
  }
  csg_f32x4 st[NLD];
  auto load_stage = [&](int s) {                 // s past the end: finite garbage or zeros, never consumed
#pragma unroll
    for (int i = 0; i < NLD; ++i) st[i] = csg_buf_load_x4(rsX, (int)goff[i], s * (WN_BK * 4), 0);
  };
  auto store_stage = [&](float* base) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      float* dst = base + loff[i];
      *(float2*)(dst) = make_float2(st[i].x, st[i].y);
      *(float2*)(dst + 2) = make_float2(st[i].z, st[i].w);
    }
  };

  // ---- this lane's tile inside each of the two tile groups, and the two input rows its wave combines
  const int j = lane & 31, h = lane >> 5;
  const int tx = j % TW, tyl = j / TW;
  // row xi of B^T d:  xi=0: d0 - d2;  1: d1 + d2;  2: d2 - d1;  3: d1 - d3
  const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  // word offsets of (row a / row b, column 0, channel pair h) for tile group 0; group 1 is 2*RPG rows further down
  const float* pa0 = smem + (2 * tyl + ia) * RS + tx * WN_PS + 2 * h;
  const float* pb0 = smem + (2 * tyl + ib) * RS + tx * WN_PS + 2 * h;

  // ---- packed weights of this wave: [xi = wave][nu][nt32][q][lane], byte offsets; q rides in the scalar offset
  unsigned uoff[4][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int nt32 = nb * 2 + nt;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
      uoff[nu][nt] = nt32 < p.NT32 ? (unsigned)((((wave * 4 + nu) * p.NT32 + nt32) * p.Q8) * 64 + lane) * 16u : CSG_OOB_OFF;
  }
  csg_f32x4 ua[4][2], ub[4][2];
  auto load_u = [&](csg_f32x4 (&u)[4][2], int q) {
    const int qq = min(q, p.Q8 - 1);             // the prefetch past the end re-reads the last block (scalar op)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) u[nu][nt] = csg_buf_load_x4(rsU, (int)uoff[nu][nt], qq * 1024, 0);
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

  // One "group" = one k-oct (8 channels) of one tile group: 16 ds_read_b64 -> 32 VALU (B^T d B) -> 32 MFMAs.
  // The loop below is software-pipelined by hand over groups: while the MFMAs of group g run, the LDS reads of
  // group g+1 are already issued and its transform is slotted between the last MFMAs (sched_group_barrier), so the
  // single wave per SIMD keeps the matrix pipe fed.
  struct RawG {
    float2 a[2][4], b[2][4];                     // [channel pair][column] of input rows ia / ib
  };
  auto read_group = [&](int bufsel, int o, int mt, RawG& r) {     // every offset below is an immediate
    const float* pa = pa0 + bufsel * BUFW + mt * (2 * RPG * RS) + 8 * o;
    const float* pb = pb0 + bufsel * BUFW + mt * (2 * RPG * RS) + 8 * o;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp)
#pragma unroll
      for (int c = 0; c < 4; ++c) {              // column c of the 4x4 patch: even/odd halves, then + c>>1 pixels
        const int co = (c & 1) * HALF + (c >> 1) * WN_PS + 4 * cp;
        r.a[cp][c] = *(const float2*)(pa + co);
        r.b[cp][c] = *(const float2*)(pb + co);
      }
  };
  auto transform = [&](const RawG& r, float2 (&v)[4][2]) {
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      float2 q[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) q[c] = make_float2(r.a[cp][c].x + sgn * r.b[cp][c].x, r.a[cp][c].y + sgn * r.b[cp][c].y);
      v[0][cp] = make_float2(q[0].x - q[2].x, q[0].y - q[2].y);
      v[1][cp] = make_float2(q[1].x + q[2].x, q[1].y + q[2].y);
      v[2][cp] = make_float2(q[2].x - q[1].x, q[2].y - q[1].y);
      v[3][cp] = make_float2(q[1].x - q[3].x, q[1].y - q[3].y);
    }
  };
  auto mfma_group = [&](int mt, const csg_f32x4 (&u)[4][2], const float2 (&v)[4][2]) {
    // k-step outermost: consecutive MFMAs go to 8 different accumulators
    // weights as the first operand: D[i = channel][j = tile] -> a lane holds one tile and runs of 4 channels
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].x, v[nu][0].x, acc[nu][mt][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].y, v[nu][0].y, acc[nu][mt][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].z, v[nu][1].x, acc[nu][mt][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
        acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].w, v[nu][1].y, acc[nu][mt][nt], 0, 0, 0);
  };
  // issue order of one pipeline step: the 16 LDS reads, 16 MFMAs back to back, then MFMA / 2 VALU alternating
#define WN_STEP_SCHED()                                   \
  __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);     \
  __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);     \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);    \
  }

  // one stage = 16 channels out of buffer `bufsel` (compile-time), entered with v0 = transform of (s, oct 0, group 0)
  RawG raw;
  float2 v0[4][2], v1[4][2];
  auto stage = [&](int s, auto bufsel_tag) {
    constexpr int bufsel = decltype(bufsel_tag)::value;
    load_stage(s + 1);
    load_u(ub, 2 * s + 1);
    __builtin_amdgcn_sched_barrier(0);           // the global loads of the next stage lead the trip

    read_group(bufsel, 0, 1, raw);
    mfma_group(0, ua, v0);
    transform(raw, v1);
    WN_STEP_SCHED();
    __builtin_amdgcn_sched_barrier(0);

    read_group(bufsel, 1, 0, raw);
    mfma_group(1, ua, v1);
    transform(raw, v0);
    WN_STEP_SCHED();
    __builtin_amdgcn_sched_barrier(0);

    load_u(ua, 2 * s + 2);
    read_group(bufsel, 1, 1, raw);
    mfma_group(0, ub, v0);
    transform(raw, v1);
    __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
    WN_STEP_SCHED();
    __builtin_amdgcn_sched_barrier(0);

    store_stage(smem + (bufsel ^ 1) * BUFW);     // the buffer every wave finished reading one trip ago
    __syncthreads();
    read_group(bufsel ^ 1, 0, 0, raw);           // first group of the next stage, under the last MFMAs of this one
    mfma_group(1, ub, v1);
    transform(raw, v0);
    WN_STEP_SCHED();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- K loop
  load_stage(0);
  load_u(ua, 0);
  store_stage(smem);
  __syncthreads();
  read_group(0, 0, 0, raw);
  transform(raw, v0);
  int s = 0;
  for (; s + 1 < p.nstage; s += 2) {
    stage(s, std::integral_constant<int, 0>());
    stage(s + 1, std::integral_constant<int, 1>());
  }
  if (s < p.nstage) stage(s, std::integral_constant<int, 0>());
  __syncthreads();                               // every wave is done reading before the epilogue reuses the LDS
  const int rows_per_group = RPG;

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
      const int ttx = jj % TW, tty = mt * rows_per_group + jj / TW;
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
            if (gate != nullptr) {
              const float4 gv = *(const float4*)(gate + pix * p.y_cs + n);
              v[0] *= gv.x > 0.f ? 1.f : p.gate_slope; v[1] *= gv.y > 0.f ? 1.f : p.gate_slope;
              v[2] *= gv.z > 0.f ? 1.f : p.gate_slope; v[3] *= gv.w > 0.f ? 1.f : p.gate_slope;
            }
            *(float4*)(y + pix * p.y_cs + n) = make_float4(v[0], v[1], v[2], v[3]);
          }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ convolution, 2 blocks per CU
// Same algorithm with HALF the tile set per block: 32 tiles (TW x TH) x 64 output channels, accumulators 4 nu x 2
// channel groups = 128 registers, so TWO blocks are resident per CU (two waves per SIMD).  A lone wave pays for every
// VMEM / LDS / VALU issue and for its prologue and epilogue with idle matrix-pipe time (k_wino_conv: 55 % busy); here
// the partner wave's MFMAs run underneath, and one block's epilogue under the other block's main loop.  To fit 256
// registers the weights are single-buffered: the eight operands of one channel group are reloaded (for the next
// k-oct) right after their last MFMA, while the other group's 16 MFMAs run.
template <int TW, int NTB = 2>     // NTB: 32-channel groups per block (1 for layers with <= 32 output channels)
__global__ __launch_bounds__(256, 2) void k_wino_conv2(WinoParams p, const float* __restrict__ x,
                                                        const float4* __restrict__ up, const float* __restrict__ bias,
                                                        const float* __restrict__ res, const float* __restrict__ gate,
                                                        float* __restrict__ y) {
  constexpr int TH = 32 / TW;
  constexpr int R = 2 * TH + 2, C = 2 * TW + 2;
  constexpr int RS = wn_row_stride(TW);
  constexpr int BUFW = R * RS + 16;
  constexpr int HALF = (C / 2) * WN_PS;
  constexpr int NLD = (R * C * 4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  int bid = wn_xcd_remap(blockIdx.x, gridDim.x);
  const int split = bid % p.ksplit;              // splits of one tile are neighbours: they share the input in L2
  bid /= p.ksplit;
  const int nb = bid % p.nblocks;
  bid /= p.nblocks;
  const int bx = bid % p.tbx;
  bid /= p.tbx;
  const int by = bid % p.tby;
  const int img = bid / p.tby;
  const int X0 = bx * 2 * TW, Y0 = by * 2 * TH;

  const csg_i32x4 rsX = csg_make_srd(x, (long long)p.B * p.H * p.W * p.x_cs * 4);
  const csg_i32x4 rsU = csg_make_srd(up, (long long)16 * p.NT32 * p.Q8 * 64 * 16);
  unsigned goff[NLD];
  int loff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = tid + 256 * i;
    goff[i] = CSG_OOB_OFF;
    loff[i] = BUFW - 16 + (tid & 3) * 4;
    if (e < R * C * 4) {
      const int pix = e >> 2, c4 = e & 3;
      const int row = pix / C, col = pix - row * C;
      const int iy = Y0 + row - 1, ix = X0 + col - 1;
      loff[i] = row * RS + ((col & 1) * (C >> 1) + (col >> 1)) * WN_PS + c4 * 4;
      if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        goff[i] = (unsigned)(((img * p.H + iy) * p.W + ix) * p.x_cs + c4 * 4) * 4u;
    }
  }
  csg_f32x4 st[NLD];
  auto load_stage = [&](int s) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) st[i] = csg_buf_load_x4(rsX, (int)goff[i], s * (WN_BK * 4), 0);
  };
  auto store_stage = [&](float* base) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      float* dst = base + loff[i];
      *(float2*)(dst) = make_float2(st[i].x, st[i].y);
      *(float2*)(dst + 2) = make_float2(st[i].z, st[i].w);
    }
  };

  const int j = lane & 31, h = lane >> 5;
  const int tx = j % TW, tyl = j / TW;
  const int ia = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
  const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const float* pa0 = smem + (2 * tyl + ia) * RS + tx * WN_PS + 2 * h;
  const float* pb0 = smem + (2 * tyl + ib) * RS + tx * WN_PS + 2 * h;

  unsigned uoff[4][NTB];
#pragma unroll
  for (int nt = 0; nt < NTB; ++nt) {
    const int nt32 = nb * NTB + nt;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
      uoff[nu][nt] = nt32 < p.NT32 ? (unsigned)((((wave * 4 + nu) * p.NT32 + nt32) * p.Q8) * 64 + lane) * 16u : CSG_OOB_OFF;
  }
  csg_f32x4 u[4][NTB];
  auto load_u_half = [&](int nt, int q) {        // operands of channel group nt for k-oct q (clamped: see k_wino_conv)
    const int qq = min(q, p.Q8 - 1);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) u[nu][nt] = csg_buf_load_x4(rsU, (int)uoff[nu][nt], qq * 1024, 0);
  };

  f32x16 acc[4][NTB];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int nt = 0; nt < NTB; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nu][nt][e] = 0.f;

  float2 v[4][2];
  auto prepare = [&](int bufsel, int o) {        // 16 ds_read_b64 + 32 VALU: V = B^T d B of this lane's tile, k-oct o
    const float* pa = pa0 + bufsel * BUFW + 8 * o;
    const float* pb = pb0 + bufsel * BUFW + 8 * o;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      float2 q[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int co = (c & 1) * HALF + (c >> 1) * WN_PS + 4 * cp;
        const float2 da = *(const float2*)(pa + co), db = *(const float2*)(pb + co);
        q[c] = make_float2(da.x + sgn * db.x, da.y + sgn * db.y);
      }
      v[0][cp] = make_float2(q[0].x - q[2].x, q[0].y - q[2].y);
      v[1][cp] = make_float2(q[1].x + q[2].x, q[1].y + q[2].y);
      v[2][cp] = make_float2(q[2].x - q[1].x, q[2].y - q[1].y);
      v[3][cp] = make_float2(q[1].x - q[3].x, q[1].y - q[3].y);
    }
  };
  auto mfma_half = [&](int nt) {                 // 16 MFMAs: k-step outermost, 4 accumulators in rotation
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].x, v[nu][0].x, acc[nu][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].y, v[nu][0].y, acc[nu][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].z, v[nu][1].x, acc[nu][nt], 0, 0, 0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) acc[nu][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[nu][nt].w, v[nu][1].y, acc[nu][nt], 0, 0, 0);
  };
  auto stage = [&](int s, auto bufsel_tag) {
    constexpr int bufsel = decltype(bufsel_tag)::value;
    load_stage(s + 1);
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      prepare(bufsel, o);
      mfma_half(0);
      __builtin_amdgcn_sched_barrier(0);
      load_u_half(0, 2 * s + o + 1);
      if (NTB > 1) {
        mfma_half(NTB - 1);
        __builtin_amdgcn_sched_barrier(0);
        load_u_half(NTB - 1, 2 * s + o + 1);
      }
    }
    store_stage(smem + (bufsel ^ 1) * BUFW);
    __syncthreads();
  };

  const int s_begin = split * p.sps, s_end = min(p.nstage, s_begin + p.sps);
  y += (long long)split * p.slab;
  load_stage(s_begin);
  load_u_half(0, 2 * s_begin);
  if (NTB > 1) load_u_half(NTB - 1, 2 * s_begin);
  store_stage(smem);
  __syncthreads();
  int s = s_begin;
  for (; s + 1 < s_end; s += 2) {
    stage(s, std::integral_constant<int, 0>());
    stage(s + 1, std::integral_constant<int, 1>());
  }
  if (s < s_end) stage(s, std::integral_constant<int, 0>());

  // ---- epilogue (as k_wino_conv, 32 tiles): R[0] = M0+M1+M2, R[1] = M1-M2-M3 per wave, A^T . through LDS
  float* rbuf = smem;                            // [xi][b][32 tiles][WN_RSE]
#pragma unroll
  for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float r0[4], r1[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m0 = acc[0][nt][4 * g + e], m1 = acc[1][nt][4 * g + e], m2 = acc[2][nt][4 * g + e], m3 = acc[3][nt][4 * g + e];
        r0[e] = m0 + m1 + m2;
        r1[e] = m1 - m2 - m3;
      }
      const int ch = 8 * g + 4 * h;
      *(float4*)(rbuf + ((wave * 2 + 0) * 32 + j) * WN_RSE + ch) = make_float4(r0[0], r0[1], r0[2], r0[3]);
      *(float4*)(rbuf + ((wave * 2 + 1) * 32 + j) * WN_RSE + ch) = make_float4(r1[0], r1[1], r1[2], r1[3]);
    }
    __syncthreads();
    {
      const int tile = tid >> 3, cq = tid & 7;   // 32 tiles x 8 channel quads: one item per thread
      const int n = nb * (32 * NTB) + nt * 32 + cq * 4;
      const int ttx = tile % TW, tty = tile / TW;
      const int oy = Y0 + 2 * tty, ox = X0 + 2 * ttx;
      if (n < p.Cout && oy < p.H && ox < p.W) {
        float4 rr[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
          for (int b = 0; b < 2; ++b) rr[xi][b] = *(const float4*)(rbuf + ((xi * 2 + b) * 32 + tile) * WN_RSE + cq * 4);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) bv = *(const float4*)(bias + n);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            float vv[4];
            const float4 q0 = rr[0][b], q1 = rr[1][b], q2 = rr[2][b], q3 = rr[3][b];
            if (a == 0) {
              vv[0] = q0.x + q1.x + q2.x; vv[1] = q0.y + q1.y + q2.y; vv[2] = q0.z + q1.z + q2.z; vv[3] = q0.w + q1.w + q2.w;
            } else {
              vv[0] = q1.x - q2.x - q3.x; vv[1] = q1.y - q2.y - q3.y; vv[2] = q1.z - q2.z - q3.z; vv[3] = q1.w - q2.w - q3.w;
            }
            vv[0] += bv.x; vv[1] += bv.y; vv[2] += bv.z; vv[3] += bv.w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (p.act == CSG_ACT_LEAKY)
                vv[e] = vv[e] > 0.f ? vv[e] : vv[e] * p.slope;
              else if (p.act == CSG_ACT_TANH)
                vv[e] = tanhf(vv[e]);
            }
            const int64_t pix = ((int64_t)img * p.H + (oy + a)) * p.W + (ox + b);
            if (res != nullptr) {
              const float4 rv = *(const float4*)(res + pix * p.y_cs + n);
              vv[0] += rv.x; vv[1] += rv.y; vv[2] += rv.z; vv[3] += rv.w;
            }
            if (gate != nullptr) {
              const float4 gv = *(const float4*)(gate + pix * p.y_cs + n);
              vv[0] *= gv.x > 0.f ? 1.f : p.gate_slope; vv[1] *= gv.y > 0.f ? 1.f : p.gate_slope;
              vv[2] *= gv.z > 0.f ? 1.f : p.gate_slope; vv[3] *= gv.w > 0.f ? 1.f : p.gate_slope;
            }
            *(float4*)(y + pix * p.y_cs + n) = make_float4(vv[0], vv[1], vv[2], vv[3]);
          }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ host side
static int wn_plan(const csg_wino_desc* d, WinoParams& p, size_t& shm, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  CSG_REQUIRE(d->H % 2 == 0 && d->W % 2 == 0 && d->W >= 8, CSG_E_UNSUPPORTED, "%s: H=%d, W=%d must be even, W >= 8", who,
              d->H, d->W);
  CSG_REQUIRE(d->Cin % 4 == 0 && d->x_cs % 4 == 0 && d->x_cs >= d->Cin && d->Cout % 4 == 0 && d->y_cs % 4 == 0 &&
                  d->y_cs >= d->Cout,
              CSG_E_UNSUPPORTED, "%s: channel counts and strides must be multiples of 4", who);
  CSG_REQUIRE((int64_t)d->B * d->H * d->W * (int64_t)(d->x_cs > d->y_cs ? d->x_cs : d->y_cs) * 4 < CSG_MAX_RECORDS, CSG_E_UNSUPPORTED,
              "%s: tensor too large for 32-bit byte offsets", who);
  CSG_REQUIRE((int64_t)16 * ((d->Cout + 31) / 32) * ((d->Cin + 7) / 8) * 1024 < CSG_MAX_RECORDS, CSG_E_UNSUPPORTED,
              "%s: packed weights too large for 32-bit byte offsets", who);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.Cout = d->Cout; p.y_cs = d->y_cs;
  static const int variant = getenv("CSG_WINO_VARIANT") ? atoi(getenv("CSG_WINO_VARIANT")) : 2;
  p.variant = variant;
  if (variant == 2) {            // 32 tiles per block, two blocks per CU
    p.TW = d->W >= 32 ? 16 : (d->W >= 16 ? 8 : 4);
    p.TH = 32 / p.TW;
  } else {                       // 64 tiles per block, one block per CU
    p.TW = d->W >= 64 ? 32 : (d->W >= 32 ? 16 : (d->W >= 16 ? 8 : 4));
    p.TH = 64 / p.TW;
  }
  p.tbx = (d->W / 2 + p.TW - 1) / p.TW;
  p.tby = (d->H / 2 + p.TH - 1) / p.TH;
  p.ntb = (variant == 2 && d->Cout <= 32) ? 1 : 2;     // 32-wide blocks for layers with <= 32 output channels
  p.nblocks = (d->Cout + 32 * p.ntb - 1) / (32 * p.ntb);
  p.RS = wn_row_stride(p.TW);
  CSG_REQUIRE(d->Cin % 16 == 0, CSG_E_UNSUPPORTED, "%s: Cin must be a multiple of 16 (one LDS stage)", who);
  p.NT32 = (d->Cout + 31) / 32;
  p.Q8 = (d->Cin + 7) / 8;
  p.act = d->act; p.slope = d->slope;
  p.nstage = (d->Cin + WN_BK - 1) / WN_BK;
  p.ksplit = 1;
  p.sps = p.nstage;
  p.slab = (long long)d->B * d->H * d->W * d->y_cs;
  const size_t in_bytes = (size_t)2 * ((2 * p.TH + 2) * p.RS + 16) * 4;        // + the dump slots of idle staging lanes
  const size_t ep_bytes = (size_t)4 * 2 * (p.variant == 2 ? 32 : 64) * WN_RSE * 4;
  shm = in_bytes > ep_bytes ? in_bytes : ep_bytes;
  return CSG_OK;
}

extern "C" {

int64_t csg_wino_pack_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return -1;
  return (int64_t)16 * cdiv(N, 32) * cdiv(K, 8) * 64 * 16;
}

int csg_wino_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                          int32_t backward_data, const float* sigma, float* packed, void* stream) {
  CSG_REQUIRE(w != nullptr && packed != nullptr && Cout > 0 && Cin > 0, CSG_E_BADSHAPE, "csg_wino_pack_weights: bad arguments");
  CSG_REQUIRE(((uintptr_t)packed % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino_pack_weights: packed must be 16-byte aligned");
  // w is the (Cout, Cin, 3, 3) weight with element strides (s_o, s_i, s_h, s_w) — contiguous or channels-last.
  // forward: n = cout, k = cin;  backward-data: n = cin, k = cout, taps flipped
  const int64_t N = backward_data ? Cin : Cout, K = backward_data ? Cout : Cin;
  const int64_t s_n = backward_data ? s_i : s_o, s_k = backward_data ? s_o : s_i;
  const int NT32 = (int)cdiv(N, 32), Q8 = (int)cdiv(K, 8);
  const int64_t total = (int64_t)16 * NT32 * Q8 * 64;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(K_WINO_PACK, (double)Cout * Cin * 9 * 4 + (double)total * 16, s);
  CSG_LAUNCH(k_wino_pack, dim3((unsigned)cdiv(Q8, 4), (unsigned)NT32), dim3(256), 0, s, w, s_n, s_k, s_h, s_w,
                     backward_data ? 1 : 0, (int)N, (int)K, sigma, NT32, Q8, (float4*)packed);
  return check_launch("csg_wino_pack_weights");
}

int csg_wino2_pack_multi_launch(const PackMulti* pm, int blocks, double bytes, hipStream_t s) {
  ProfScope ps(K_WINO_PACK, bytes, s);
  CSG_LAUNCH(k_wino_pack_multi, dim3((unsigned)blocks), dim3(256), 0, s, *pm);
  return check_launch("csg_wino_pack_weights_multi(F(2x2,3x3))");
}

// Split over the input channels when the tile grid alone cannot fill the chip (backward-data of the gamma/beta
// convolutions: 128 output channels, 2048 input channels): only without an epilogue (bias / activation / residual)
// and with a dense output, the slabs are summed in a fixed order by k_slab_reduce.
static void wn_split_plan(WinoParams& p, bool plain) {
  if (p.variant != 2 || !plain || p.y_cs != p.Cout) return;
  const int64_t blocks = (int64_t)p.B * p.tby * p.tbx * p.nblocks;
  if (blocks >= 384 || p.nstage < 16) return;
  int ks = (int)((512 + blocks - 1) / blocks);
  if (ks > p.nstage / 8) ks = p.nstage / 8;
  if (ks < 2) return;
  p.sps = (p.nstage + ks - 1) / ks;
  p.ksplit = (p.nstage + p.sps - 1) / p.sps;
}

int64_t csg_wino_conv_workspace(const csg_wino_desc* d) {
  WinoParams p;
  size_t shm = 0;
  if (wn_plan(d, p, shm, "csg_wino_conv_workspace")) return -1;
  wn_split_plan(p, d->act == CSG_ACT_NONE);
  return p.ksplit > 1 ? (int64_t)p.ksplit * p.slab * 4 : 0;
}

int csg_wino_conv(const csg_wino_desc* d, const float* x, const float* packed, const float* bias, const float* residual,
                  const float* gate, float gate_slope, float* y, float* workspace, int64_t workspace_bytes, void* stream) {
  WinoParams p;
  size_t shm = 0;
  int rc = wn_plan(d, p, shm, "csg_wino_conv");
  if (rc) return rc;
  wn_split_plan(p, d->act == CSG_ACT_NONE && bias == nullptr && residual == nullptr && gate == nullptr);
  p.gate_slope = gate_slope;
  CSG_REQUIRE(gate == nullptr || ((uintptr_t)gate % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino_conv: gate must be 16-byte aligned");
  if (p.ksplit > 1 && (workspace == nullptr || workspace_bytes < (int64_t)p.ksplit * p.slab * 4)) {   // no slabs: unsplit
    p.ksplit = 1;
    p.sps = p.nstage;
  }
  float* const y_final = y;
  if (p.ksplit > 1) {
    CSG_REQUIRE(((uintptr_t)workspace % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino_conv: workspace must be 16-byte aligned");
    y = workspace;
  }
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_wino_conv: pointers must be 16-byte aligned");
  static DeviceOnce attr_once;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_once.pending(dev)) {
    const void* fns[10] = {(const void*)k_wino_conv<32>, (const void*)k_wino_conv<16>, (const void*)k_wino_conv<8>,
                          (const void*)k_wino_conv<4>, (const void*)k_wino_conv2<16, 2>, (const void*)k_wino_conv2<8, 2>,
                          (const void*)k_wino_conv2<4, 2>, (const void*)k_wino_conv2<16, 1>, (const void*)k_wino_conv2<8, 1>,
                          (const void*)k_wino_conv2<4, 1>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_wino_conv: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    attr_once.mark(dev);
  }
  CSG_REQUIRE(shm <= 96 * 1024, CSG_E_UNSUPPORTED, "csg_wino_conv: %zu bytes of LDS", shm);
  hipStream_t s = (hipStream_t)stream;
  const int64_t grid = (int64_t)p.B * p.tby * p.tbx * p.nblocks * p.ksplit;
  CSG_REQUIRE(grid < (1ll << 31), CSG_E_UNSUPPORTED, "csg_wino_conv: grid too large");
  // algorithmic FLOPs of the DIRECT convolution this replaces (2 * M * 9*Cin * Cout): what FlopCounterMode counts
  ProfScope ps(K_WINO_CONV, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, s);
  const float4* up = (const float4*)packed;
  if (p.variant == 2) {
    if (p.ntb == 1) {
      if (p.TW == 16)
        CSG_LAUNCH((k_wino_conv2<16, 1>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
      else if (p.TW == 8)
        CSG_LAUNCH((k_wino_conv2<8, 1>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
      else
        CSG_LAUNCH((k_wino_conv2<4, 1>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
    } else if (p.TW == 16)
      CSG_LAUNCH((k_wino_conv2<16, 2>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
    else if (p.TW == 8)
      CSG_LAUNCH((k_wino_conv2<8, 2>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
    else
      CSG_LAUNCH((k_wino_conv2<4, 2>), dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
    rc = check_launch("csg_wino_conv");
    if (rc == CSG_OK && p.ksplit > 1) {
      launch_slab_reduce(workspace, p.slab, y_final, nullptr, 0, nullptr, p.ksplit, s);
      rc = check_launch("csg_wino_conv(slab sum)");
    }
    return rc;
  }
  if (p.TW == 32)
    CSG_LAUNCH(k_wino_conv<32>, dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
  else if (p.TW == 16)
    CSG_LAUNCH(k_wino_conv<16>, dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
  else if (p.TW == 8)
    CSG_LAUNCH(k_wino_conv<8>, dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
  else
    CSG_LAUNCH(k_wino_conv<4>, dim3((unsigned)grid), dim3(256), shm, s, p, x, up, bias, residual, gate, y);
  return check_launch("csg_wino_conv");
}

}  // extern "C"

// ====================================================================================== weight gradient
// dW (3x3) of a 3x3 / stride 1 / pad 1 convolution by Winograd F(3x3, 2x2): per 2x2 tile of dY and the 4x4 input
// patch around it,
//     dW += A^T [ (G dY_t G^T) (.) (B^T X_t B) ] A            16 multiplications instead of 36
// with A^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,1]], G = [[1,0],[1/2,1/2],[1/2,-1/2],[0,1]] and
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,-1,0,1]].  The sum over tiles is taken INSIDE the brackets: 16 GEMMs
//     Acc_p[cout][cin] = sum_tiles E_p[tile][cout] * V_p[tile][cin]
// whose reduction index is the tile.  A block owns 64 cout x 64 cin x 16 positions over a slice of the tiles; wave w
// owns row xi = w.  Both operands are formed in registers from plain coalesced global loads (32 consecutive channels
// per half-wave; the two half-waves work on two consecutive tiles = the k-pair of v_mfma_f32_32x32x2_f32): no LDS and
// no barrier in the main loop.  Slabs per tile slice + the ordered reduction of igemm.hip (k_wgrad_reduce) give a
// bit-reproducible result; the bias gradient rides along in wave 1, which loads all four pixels of every dY tile.
struct WinoWgParams {
  int B, H, W, Cin, x_cs, Cout, y_cs;
  int RXn, RYn, nregions;        // stage regions per tile row / column of an image, in all
  int cblocks, kblocks;
  int nsplit, rps;               // regions per split
  int nt;                        // input-channel groups of 32 per block (2: one block per CU, 1: two)
};

#define WG_EPS 33                // words per row of the epilogue exchange buffer

// Stage = TSX x TSY = 16 tiles (8 k-pairs of v_mfma_f32_32x32x2_f32) of one image: its (2TSY+2) x (2TSX+2) input
// patch and 2TSY x 2TSX dY pixels, 64 channels each, are staged in LDS by 16-byte global loads (13 per thread and
// stage; the first version fetched every operand with its own dword load — 24 VMEM instructions per 16 MFMAs, and one
// VMEM issue costs the lone wave of a SIMD ~40 matrix-pipe cycles).  Operands are then read with ds_read_b32 (32
// consecutive channels per half-wave: conflict-free), transformed in registers (G dY G^T without its 1/2 factors,
// which are folded into the epilogue; B^T X B) and fed to 16 MFMAs per k-pair.  Two LDS buffers, global loads of stage
// s+1 issued before the MFMAs of stage s, one barrier per stage.  Regions tile the image exactly (host-checked), so
// only the one-pixel halo needs masking: per-thread flags x uniform edge conditions, one v_cndmask per load; the
// region's position rides in the scalar offset of the buffer loads.
template <int TSX, int NT>
__global__ __launch_bounds__(256, NT == 2 ? 1 : 2) void k_wino_wgrad(WinoWgParams p, const float* __restrict__ x,
                                                        const float* __restrict__ dy, float* __restrict__ slabs,
                                                        float* __restrict__ dbslabs) {
  constexpr int TSY = 16 / TSX;
  constexpr int XR = 2 * TSY + 2, XC = 2 * TSX + 2, YR = 2 * TSY, YC = 2 * TSX;
  constexpr int XCH = 32 * NT, XF4 = XCH / 4;    // input channels per block and pixel, as floats / as 16-byte pieces
  constexpr int NLX = (XR * XC * XF4 + 255) / 256, NLY = (YR * YC * 16) / 256;
  static_assert((YR * YC * 16) % 256 == 0, "dY staging divides evenly");
  constexpr int XW = NLX * 256 * 4, YW = YR * YC * 64, BUFW = XW + YW;     // words; the X area is padded to whole DMAs
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  int bid = wn_xcd_remap(blockIdx.x, gridDim.x);
  // all (cout block, cin block) pairs of one slice are adjacent: the slice's dY and X stay in that XCD's L2
  const int kb = bid % p.kblocks;                // input-channel block of XCH channels
  bid /= p.kblocks;
  const int cb = bid % p.cblocks;
  const int sp = bid / p.cblocks;
  const int r0 = sp * p.rps, r1 = min(p.nregions, r0 + p.rps);

  // rows of the dY tile / of the input patch this wave combines (unscaled: the 1/2 of G are applied in the epilogue)
  //   e = G dY:  xi=0: dY0;  1: dY0+dY1;  2: dY0-dY1;  3: dY1          r = B^T X: xi=0: X0-X2; 1: X1+X2; 2: X2-X1; 3: X3-X1
  const int ia = wave == 0 ? 0 : (wave == 1 ? 1 : (wave == 2 ? 2 : 3));
  const int ib = wave == 0 ? 2 : (wave == 1 ? 2 : 1);
  const float sgn = wave == 1 ? 1.0f : -1.0f;
  const float ea = wave == 3 ? 0.f : 1.f;
  const float eb = wave == 0 ? 0.f : (wave == 2 ? -1.f : 1.f);

  // ---- staging plan (k-invariant).  Staging is LDS-DMA (buffer_load_dwordx4 ... lds): element e = tid + 256 i of
  // the linear [pixel][64 channels] image lands at LDS byte 16 e — exactly the wave-uniform base + 16 * lane the
  // instruction writes — so the stage costs no VGPRs and no ds_write; out-of-range lanes (halo outside the image,
  // channel tail, padding of the last DMA) are written as zeros.  Relative byte offsets inside the region (the X
  // descriptor sits one row + one pixel before the tensor, so halo offsets are non-negative) and halo flags:
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x - (int64_t)(p.W + 1) * p.x_cs), 0, (int)(((long long)p.B * p.H * p.W + p.W + 1) * p.x_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long long)p.B * p.H * p.W * p.y_cs * 4), 0x00020000);
  unsigned xoff[NLX], yoff[NLY];
  int xflag[NLX];                                // flag bits: 1 first row, 2 last row, 4 first col, 8 last col
#pragma unroll
  for (int i = 0; i < NLX; ++i) {
    const int e = tid + 256 * i;
    const int pix = e / XF4, c4 = e - pix * XF4;
    const int row = pix / XC, col = pix - row * XC;
    const int ch = kb * XCH + c4 * 4;
    const bool ok = (e < XR * XC * XF4) & (ch < p.Cin);
    xoff[i] = ok ? (unsigned)((row * p.W + col) * p.x_cs + ch) * 4u : CSG_OOB_OFF;
    xflag[i] = (row == 0 ? 1 : 0) | (row == XR - 1 ? 2 : 0) | (col == 0 ? 4 : 0) | (col == XC - 1 ? 8 : 0);
  }
#pragma unroll
  for (int i = 0; i < NLY; ++i) {
    const int e = tid + 256 * i;
    const int pix = e >> 4, c4 = e & 15;
    const int row = pix / YC, col = pix - row * YC;
    const int ch = cb * 64 + c4 * 4;
    yoff[i] = ch < p.Cout ? (unsigned)((row * p.W + col) * p.y_cs + ch) * 4u : CSG_OOB_OFF;
  }
  typedef __attribute__((address_space(3))) void* lds_ptr;
  auto dma_stage = [&](int r, int bufsel) {      // region r -> (image, region row, region column); r >= r1: not consumed
    const int rr = min(r, p.nregions - 1);
    const int rx = rr % p.RXn, t = rr / p.RXn;
    const int ry = t % p.RYn, img = t / p.RYn;
    const int y0 = ry * 2 * TSY, x0 = rx * 2 * TSX;
    const int edge = (y0 == 0 ? 1 : 0) | (y0 + 2 * TSY == p.H ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + 2 * TSX == p.W ? 8 : 0);
    const int pix0 = (img * p.H + y0) * p.W + x0;
    float* base = smem + bufsel * BUFW + wave * 256;       // this wave's 64 x 16 bytes of every 256-lane DMA row
#pragma unroll
    for (int i = 0; i < NLX; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(base + i * 1024), 16,
                                               (int)((xflag[i] & edge) ? CSG_OOB_OFF : xoff[i]), pix0 * p.x_cs * 4, 0, 0);
#pragma unroll
    for (int i = 0; i < NLY; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr)(base + XW + i * 1024), 16, (int)yoff[i], pix0 * p.y_cs * 4, 0, 0);
  };

  const bool do_db = dbslabs != nullptr && kb == 0 && wave == 1;
  float dbacc[2] = {0.f, 0.f};

  f32x16 acc[4][2][NT];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[nu][mt][nt][e] = 0.f;

  struct Raw {
    float dy[2][2][2];    // [cout group][row][col]
    float xr[NT][2][4];   // [cin group][row a/b][col]
  };
  // tile t = 2*kp + h of the stage: (tsx, tsy) = (t % TSX, t / TSX); every offset is an immediate once kp is unrolled
  // h = 1: the next tile, two pixels to the right (TSX is even)
  const float* lxx = smem + h * (2 * XCH) + c;
  const float* lxy = smem + XW + h * (2 * 64) + c;
  auto read_pair = [&](int bufsel, int kp, Raw& r) {
    const int t = 2 * kp, tsx = t % TSX, tsy = t / TSX;
    const float* bx = lxx + bufsel * BUFW;
    const float* by = lxy + bufsel * BUFW;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jx = 0; jx < 2; ++jx) r.dy[m][i][jx] = by[((2 * tsy + i) * YC + 2 * tsx + jx) * 64 + m * 32];
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) {
        r.xr[m][0][jx] = bx[((2 * tsy + ia) * XC + 2 * tsx + jx) * XCH + m * 32];
        r.xr[m][1][jx] = bx[((2 * tsy + ib) * XC + 2 * tsx + jx) * XCH + m * 32];
      }
  };
  auto compute_pair = [&](const Raw& r) {
    float E[4][2], V[4][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float e0 = ea * r.dy[m][0][0] + eb * r.dy[m][1][0];
      const float e1 = ea * r.dy[m][0][1] + eb * r.dy[m][1][1];
      E[0][m] = e0;
      E[1][m] = e0 + e1;
      E[2][m] = e0 - e1;
      E[3][m] = e1;
      if (do_db) dbacc[m] += (r.dy[m][0][0] + r.dy[m][0][1]) + (r.dy[m][1][0] + r.dy[m][1][1]);
    }
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      float q[4];
#pragma unroll
      for (int jx = 0; jx < 4; ++jx) q[jx] = r.xr[m][0][jx] + sgn * r.xr[m][1][jx];
      V[0][m] = q[0] - q[2];
      V[1][m] = q[1] + q[2];
      V[2][m] = q[2] - q[1];
      V[3][m] = q[3] - q[1];
    }
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nu][mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(E[nu][mt], V[nu][nt], acc[nu][mt][nt], 0, 0, 0);
  };

  // one stage out of buffer `bufsel` (compile-time): reads of pair kp+1 are issued before the MFMAs of pair kp
  Raw ra, rb;
  auto stage = [&](int r, auto bufsel_tag) {
    constexpr int bufsel = decltype(bufsel_tag)::value;
    dma_stage(r + 1, bufsel ^ 1);                // the buffer every wave finished reading before the last barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kp = 0; kp < 8; kp += 2) {
      read_pair(bufsel, kp + 1, rb);
      compute_pair(ra);
      __builtin_amdgcn_sched_barrier(0);         // keeps the LDS reads one pair ahead, not eight (registers)
      if (kp + 2 < 8) {
        read_pair(bufsel, kp + 2, ra);
        compute_pair(rb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();                             // (its fence waits for this wave's DMAs: vmcnt(0))
    read_pair(bufsel ^ 1, 0, ra);                // first pair of the next stage under the last MFMAs of this one
    compute_pair(rb);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (r0 < r1) {
    dma_stage(r0, 0);
    __syncthreads();
    read_pair(0, 0, ra);
    int r = r0;
    for (; r + 1 < r1; r += 2) {
      stage(r, std::integral_constant<int, 0>());
      stage(r + 1, std::integral_constant<int, 1>());
    }
    if (r < r1) stage(r, std::integral_constant<int, 0>());
  }
  __syncthreads();

  // ---- epilogue: with g = (1, 1/2, 1/2, 1) the true accumulators are g_xi g_nu Acc'.  Per wave
  //      R_xi[b] = g_xi sum_nu g_nu Acc'[xi][nu] A^T[b][nu];  dW[a][b] = sum_xi A^T[a][xi] R_xi[b] through LDS,
  //      one 32x32 (cout, cin) quadrant at a time; slab layout [split][Cout][tap = 3a+b][Cin]
  const float gx = (wave == 1 || wave == 2) ? 0.5f : 1.0f;
  float* slab = slabs + (int64_t)sp * p.Cout * 9 * p.Cin;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float a0 = acc[0][mt][nt][e], a1 = 0.5f * acc[1][mt][nt][e], a2 = 0.5f * acc[2][mt][nt][e], a3 = acc[3][mt][nt][e];
        const int i = (e & 3) + 8 * (e >> 2) + 4 * h;             // cout row inside the quadrant
        smem[((wave * 3 + 0) * 32 + i) * WG_EPS + c] = gx * (a0 + a1 + a2);
        smem[((wave * 3 + 1) * 32 + i) * WG_EPS + c] = gx * (a1 - a2);
        smem[((wave * 3 + 2) * 32 + i) * WG_EPS + c] = gx * (a1 + a2 + a3);
      }
      __syncthreads();
      const int kcol = tid & 31;
      const int cin_g = kb * XCH + nt * 32 + kcol;
#pragma unroll
      for (int rep = 0; rep < 4; ++rep) {
        const int i = (tid >> 5) + 8 * rep;
        const int cout_g = cb * 64 + mt * 32 + i;
        if (cout_g < p.Cout && cin_g < p.Cin) {
          float R[4][3];
#pragma unroll
          for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int b = 0; b < 3; ++b) R[xi][b] = smem[((xi * 3 + b) * 32 + i) * WG_EPS + kcol];
          float* dst = slab + (int64_t)cout_g * 9 * p.Cin + cin_g;
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            dst[(0 * 3 + b) * p.Cin] = R[0][b] + R[1][b] + R[2][b];
            dst[(1 * 3 + b) * p.Cin] = R[1][b] - R[2][b];
            dst[(2 * 3 + b) * p.Cin] = R[1][b] + R[2][b] + R[3][b];
          }
        }
      }
      __syncthreads();
    }
  if (do_db) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float tot = dbacc[m] + __shfl_xor(dbacc[m], 32, 64);
      const int co = cb * 64 + m * 32 + c;
      if (h == 0 && co < p.Cout) dbslabs[(int64_t)sp * p.Cout + co] = tot;
    }
  }
}

static int wn_wg_tsx(int W) { return W >= 32 ? 16 : (W >= 16 ? 8 : 4); }

static int wn_wg_plan(const csg_wino_desc* d, WinoWgParams& p, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  const int tsx = wn_wg_tsx(d->W), tsy = 16 / tsx;
  CSG_REQUIRE(d->W >= 8 && d->W % (2 * tsx) == 0 && d->H % (2 * tsy) == 0, CSG_E_UNSUPPORTED,
              "%s: H=%d, W=%d must be multiples of the %dx%d-tile stage", who, d->H, d->W, tsx, tsy);
  CSG_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0 && d->x_cs % 4 == 0 && d->y_cs % 4 == 0 && d->x_cs >= d->Cin &&
                  d->y_cs >= d->Cout,
              CSG_E_UNSUPPORTED, "%s: channel counts and strides must be multiples of 4", who);
  CSG_REQUIRE(((int64_t)d->B * d->H * d->W + d->W + 1) * (int64_t)(d->x_cs > d->y_cs ? d->x_cs : d->y_cs) * 4 < CSG_MAX_RECORDS,
              CSG_E_UNSUPPORTED, "%s: tensor too large for 32-bit byte offsets", who);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.Cout = d->Cout; p.y_cs = d->y_cs;
  p.RXn = d->W / (2 * tsx);
  p.RYn = d->H / (2 * tsy);
  const int64_t nr = (int64_t)d->B * p.RXn * p.RYn;
  CSG_REQUIRE(nr < (1ll << 30), CSG_E_UNSUPPORTED, "%s: too many regions", who);
  p.nregions = (int)nr;
  static const int variant = getenv("CSG_WINO_WGRAD_VARIANT") ? atoi(getenv("CSG_WINO_WGRAD_VARIANT")) : 2;
  p.nt = variant == 2 ? 1 : 2;                   // 32 (two blocks per CU) or 64 input channels per block
  p.cblocks = (d->Cout + 63) / 64;
  p.kblocks = (d->Cin + 32 * p.nt - 1) / (32 * p.nt);
  // 1 or 2 blocks per CU are resident: aim at ~2 waves of blocks, at least 8 stages per block, at most 512 slabs
  const int tiles2d = p.cblocks * p.kblocks;
  int ns = (512 * (3 - p.nt) + tiles2d - 1) / tiles2d;
  const int max_ns = (int)((nr + 7) / 8);
  if (ns > max_ns) ns = max_ns;
  if (ns > 512) ns = 512;
  if (ns < 1) ns = 1;
  p.rps = (int)((nr + ns - 1) / ns);
  p.nsplit = (int)((nr + p.rps - 1) / p.rps);
  return CSG_OK;
}

extern "C" {

int64_t csg_wino_bwd_weight_workspace(const csg_wino_desc* d) {
  WinoWgParams p;
  if (wn_wg_plan(d, p, "csg_wino_bwd_weight_workspace")) return -1;
  return (int64_t)p.nsplit * d->Cout * (9 * (int64_t)d->Cin + 1) * 4;
}

int csg_wino_bwd_weight(const csg_wino_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                        int64_t workspace_bytes, void* stream) {
  WinoWgParams p;
  int rc = wn_wg_plan(d, p, "csg_wino_bwd_weight");
  if (rc) return rc;
  const int64_t wsize = (int64_t)d->Cout * 9 * d->Cin;
  const int64_t need = (int64_t)p.nsplit * (wsize + d->Cout) * 4;
  CSG_REQUIRE(workspace != nullptr && workspace_bytes >= need, CSG_E_WORKSPACE, "csg_wino_bwd_weight: workspace %ld < %ld bytes",
              (long)workspace_bytes, (long)need);
  static DeviceOnce attr_once;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int tsx = wn_wg_tsx(d->W), tsy = 16 / tsx;
  // per buffer: the X patch (32*nt channels) padded to whole 256-lane DMAs + the dY pixels (64 channels)
  const size_t x_f4 = (size_t)(((2 * tsy + 2) * (2 * tsx + 2) * 8 * p.nt + 255) / 256) * 256;
  const size_t stage_bytes = 2 * (x_f4 * 16 + (size_t)(2 * tsy) * (2 * tsx) * 64 * 4);
  const size_t ep_bytes = (size_t)4 * 3 * 32 * WG_EPS * 4;
  const size_t shm = stage_bytes > ep_bytes ? stage_bytes : ep_bytes;
  if (attr_once.pending(dev)) {
    const void* fns[6] = {(const void*)k_wino_wgrad<16, 2>, (const void*)k_wino_wgrad<8, 2>, (const void*)k_wino_wgrad<4, 2>,
                          (const void*)k_wino_wgrad<16, 1>, (const void*)k_wino_wgrad<8, 1>, (const void*)k_wino_wgrad<4, 1>};
    for (const void* fn : fns) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
      CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_wino_bwd_weight: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    attr_once.mark(dev);
  }
  CSG_REQUIRE(shm <= 128 * 1024, CSG_E_UNSUPPORTED, "csg_wino_bwd_weight: %zu bytes of LDS", shm);
  hipStream_t s = (hipStream_t)stream;
  float* dbslabs = db != nullptr ? workspace + (int64_t)p.nsplit * wsize : nullptr;
  if (p.nsplit == 1) {          // a single slice: the block results ARE the gradient
    workspace = dw;
    dbslabs = db;
  }
  {
    ProfScope ps(K_WINO_WGRAD, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, s);
    const dim3 grid((unsigned)(p.cblocks * p.kblocks * p.nsplit));
    if (p.nt == 2) {
      if (tsx == 16)
        CSG_LAUNCH((k_wino_wgrad<16, 2>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
      else if (tsx == 8)
        CSG_LAUNCH((k_wino_wgrad<8, 2>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
      else
        CSG_LAUNCH((k_wino_wgrad<4, 2>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
    } else {
      if (tsx == 16)
        CSG_LAUNCH((k_wino_wgrad<16, 1>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
      else if (tsx == 8)
        CSG_LAUNCH((k_wino_wgrad<8, 1>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
      else
        CSG_LAUNCH((k_wino_wgrad<4, 1>), grid, dim3(256), shm, s, p, x, dy, workspace, dbslabs);
    }
    rc = check_launch("csg_wino_bwd_weight");
    if (rc) return rc;
  }
  if (p.nsplit > 1) {
    ProfScope ps(K_WGRAD_REDUCE, (double)(p.nsplit + 1) * wsize * 4, s);
    launch_slab_reduce(workspace, wsize, dw, dbslabs, db != nullptr ? d->Cout : 0, db, p.nsplit, s);
    rc = check_launch("csg_wino_bwd_weight(reduce)");
  }
  return rc;
}

}  // extern "C"
