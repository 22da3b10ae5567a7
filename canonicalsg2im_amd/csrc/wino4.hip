// K8w4 — 3x3 / stride 1 / pad 1 convolution by Winograd F(4x4,3x3) on the fp32 matrix cores of gfx950.
//
// Same call sites as wino.hip (generator.py:28; architecture.py:29-31; normalization.py:89-94 and their backward-data
// passes) on maps at least 32 pixels wide: a 4x4 output tile from a 6x6 input tile with 36 multiplications per
// (cin, cout) pair instead of 144 — 2.25 per output where F(2x2,3x3) needs 4 and the direct sum 9.
//
//     U = G g G^T   (6x6 per (cout,cin), packed once per weight version by k_wino4_pack)
//     V = B^T d B   (6x6 per tile and channel, formed in registers right before the MFMAs)
//     M_p = sum_cin V_p U_p  for the 36 positions p = (xi, nu)   <- 36 independent GEMMs on v_mfma_f32_32x32x2_f32
//     Y = A^T M A   (4x4 outputs; bias / activation / residual / gate fused behind it)
//
// Interpolation points {0, 1, -1, 1/2, -2, inf}: among the 5-point sets tried on the host (fp32 transforms, fp32
// k-ordered accumulation, against the fp64 direct sum: tools/wino_error_model.py) this one has the smallest error —
// 2.2e-6 of the output scale at Cin = 128 and 5.5e-6 at Cin = 1024 (max over the outputs; direct fp32 0.9e-6 / 2.3e-6,
// F(2x2,3x3) 0.4e-6 / 1.3e-6; the textbook {0, +-1, +-2} reads 6.3e-6 / 1.4e-5).  tests/test_gpu_wino4.py holds the
// kernel to < 1e-5 of the output scale against fp64.
//
// Work decomposition.  A block owns 32 tiles (8 x 4: 32 x 16 output pixels of one image) x 64 output channels x all 36
// positions with TWELVE waves, one block per CU (three waves per SIMD, placed evenly by construction).
//   * MFMA role: wave w owns row xi = w mod 6 of channel group w / 6: six accumulators of 32x32 (96 registers).
//   * B operand (U): as in wino.hip every (position, channel group) is consumed by exactly one wave, so U never goes
//     through LDS: MFMA operand order [xi][nu][cout/32][cin/8][lane][4], one 16-byte load per four MFMAs, L2 resident,
//     fetched three positions ahead through a ring of three registers quads.
//   * A operand (V): the raw input region (18 x 34 pixels x 8 channels per stage) is staged in LDS with the columns
//     de-interleaved modulo 4 and the row groups skewed by two words (word(y, x, ch) = 288 y + 2 (y >> 2) +
//     8 ((x & 3) * 9 + (x >> 2)) + ch): every ds_read_b64 of a half-wave (8 x 4 tiles) hits 32 different bank pairs.
//     Transform role: wave w forms row xi = w mod 6 of V = B^T d B for the channel pairs 2 (w / 6) + {0, 1} (one per
//     half-wave) — five input rows with wave-uniform coefficients (scalar registers), then the column transform with
//     compile-time coefficients, ~96 VALU per wave and stage — and writes it to LDS in MFMA operand order
//     [xi][nu][half][tile][4]; one stage later every wave reads the six operands of its row with one conflict-free
//     ds_read_b128 each (four MFMAs per read).  The transform of a stage is computed ONCE per block.
//   * K loop: 8 channels per stage (24 MFMAs per wave).  Stage k: V[k+1] from raw[k+1]; MFMAs out of V[k]; raw[k+2]
//     (global loads issued a stage earlier) into the buffer raw[k] left; one barrier.
//   * Epilogue: each wave reduces its row over nu (M A), the six rows meet in LDS (A^T .) two output columns at a
//     time, 16-byte stores.
//
// How it got here (MI355X, B = 16; profiles/archive/r03_wino4_notes.md).  (1) Six-wave blocks, two per CU, V formed in
// registers by every wave: the dispatcher never co-scheduled two six-wave workgroups (a six-wave group lands 2+2+1+1 on
// the four SIMDs and the next one needs the mirror image): half occupancy, 31 % matrix-pipe busy, 1.4x SLOWER than
// F(2x2,3x3).  (2) Twelve-wave blocks (two channel groups): full occupancy, par with F(2x2,3x3).  Ablations of that
// kernel on 512 -> 256 channels at 64 x 64: everything 0.64 ms; without the LDS reads of the transform 0.43; without
// MFMAs 0.34; MFMAs + column transform only 0.38; nothing but prologue + epilogue 0.09 — the transform and the MFMAs
// ADD UP, also when the row transform is software-pipelined one half-stage ahead into a second register set, and a
// half-stage stagger of waves 4-7 (MI355X_MICROARCH.md, two waves per SIMD, item 9) buys 1-4 %: on this chip the fp32
// MFMA runs at the fp32 VALU rate and VALU work of the same SIMD does not hide under it.  (3) Hence this form: the
// transform is computed once per block instead of once per channel group (~4 instead of ~8 VALU per MFMA): 1.19-1.40x
// FASTER than F(2x2,3x3) on the generator's shapes (128 -> 256 channels at 256 x 256: 2.07 vs 2.46 ms).
// (4) Round 5: on large grids the kernel runs as ONE block per CU that walks its items with the stage pipeline carried
// across them (k_wino4_conv_v<4, true>, below): what a block paid outside its loop on a Cin = 128 layer — 8-9 k cycles
// until the first stage lands, 2.5 k to prime, of 122 k — shrinks to the transform of the first stage; 1.03-1.06x on the
// 256 x 256 and 128 x 128 layers, 1.13x at Cin = 32, bit-identical outputs.  Two things had made the same idea lose in
// round 3: loads issued on some paths only (the compiler's vmcnt waits then also cover the staging loads issued a moment
// earlier: +6 % per stage), and values hoisted out of / sunk into the item loop that spilled around the main loop.
// Two details that cost 10 %: a volatile LDS read through a GENERIC pointer becomes a FLAT load with a 64-bit address
// of its own (w4_lds_cv2 below), and hipcc merges neighbouring ds_read_b64 into ds_read2_b64, which is served on a
// 32-bank map in 16-lane groups where the skewed layout is 2-way conflicted.
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "csg_buffer.h"
#include "csg_common.h"
#include "csg_pack.h"
#include "csg_reduce.h"

using namespace csg;

typedef __attribute__((ext_vector_type(16))) float f32x16;
// LDS reads that must stay single ds_read_b64 instructions: volatile (no ds_read2 merging) THROUGH an address-space-3
// pointer — a volatile access through a generic pointer is not rewritten to LDS by the compiler and becomes a FLAT load
// with a 64-bit address of its own (dozens of address pairs, spilled)
typedef const volatile __attribute__((address_space(3))) csg_f32x2* w4_lds_cv2;

#define W4_TW 8
#define W4_TH 4
#define W4_CQ 9                   // columns per class (x & 3): ceil(34 / 4)
#define W4_PS 8                   // words per staged pixel = channels per stage
#define W4_RSE 36                 // words per tile row of the epilogue exchange buffer (32 channels + 4)
#define W4_NLD 2                  // float4 global loads per thread and stage (18 * 34 * 2 / 768 rounded up)
#define W4_THREADS 768             // twelve waves: two channel groups x six rows xi
#define W4_RBUF (6 * 2 * 32 * W4_RSE)       // words of one channel group's epilogue exchange buffer

// Developer build (-DW4_TRACE, tools/wino4_trace.py): thread 0 of the first 8192 blocks leaves shader-clock timestamps
// at the phase boundaries of the kernel; csg_wino4_trace_read copies them out.
#ifdef W4_TRACE
__device__ unsigned long long w4_trace[8192 * 8];
__device__ unsigned long long w4_trace2[8192 * 8];   // per item (persistent) / per block: stage-level markers
#define W4_T(i)                                                                                   \
  if (threadIdx.x == 0 && blockIdx.x < 8192) w4_trace[blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();
// persistent form: the same table indexed by ITEM (markers 0 item start | 3 main loop done | 6 epilogue done | 7 V[0] of the next item)
#define W4_TI(i)                                                                                   \
  if (threadIdx.x == 0 && (P ? v : (int)blockIdx.x) < 8192)                                        \
    w4_trace2[(P ? v : (int)blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter();
#else
#define W4_T(i)
#define W4_TI(i)
#endif

struct Wino4Params {
  int B, H, W, Cin, x_cs, Cout, y_cs;
  int Ho, Wo, pad;     // output size and zero padding (F(4x4,3x3): Ho = H, Wo = W, pad = 1)
  int tbx, tby;        // block regions per image
  int nblocks;         // ceil(Cout / 64)
  int NT32, Q8;        // extents of the packed weights
  int act;
  float slope;
  float gate_slope;
  int nt_off;          // first 32-channel tile of the packed operand this launch computes (a slice of the outputs)
  int mod, g_cs;       // SPADE modulation in the epilogue (below: 1 = beta half, 2 = joint); pixel stride of the gamma tensor
  int gb_off;          // joint gamma | beta launch: tiles of the packed operand between a gamma tile and its beta tile (C / 32)
  float mod_slope;
  const float* mod_mean;
  const float* mod_invstd;
  int nstage;          // Cin / 8
  int ksplit, sps;     // input-channel stages cut into ksplit ranges of sps stages, one output slab each
  long long slab;      // floats per slab (B*H*W*y_cs)
  int nitems;          // persistent kernel: (image, region, channel block) items of the launch
};

__device__ __forceinline__ int w4_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// B^T of F(4,3) on the points {0, 1, -1, 1/2, -2, inf} (rows = xi, columns = input rows d0..d5):
//   [ 1  -1.5  -2    1.5   1    0 ]      [ 0  -1    0.5   2.5   1   0 ]      [ 0   1   -2.5   0.5   1   0 ]
//   [ 0  -2    -1    2     1    0 ]      [ 0   0.5  -1   -0.5   1   0 ]      [ 0   1   -1.5  -2     1.5 1 ]
// row xi touches input rows rb .. rb+4 with rb = 0 for xi = 0, else 1 (rows 1..4 only rb .. rb+3)

// ------------------------------------------------------------------------------------ weight packing
// up[(((xi*6+nu)*NT32 + nt)*Q8 + q)*64 + lane] (float4) = U[xi][nu][n = nt*32 + (lane&31)][k], k = 8q + 2h + {0,1}
// (.x,.y) and 8q + 4 + 2h + {0,1} (.z,.w), h = lane>>5; zero beyond N / K — the layout of k_wino_pack with 36
// positions.  G = [[1,0,0],[1/3,1/3,1/3],[-1/3,1/3,-1/3],[-16/15,-8/15,-4/15],[1/15,-2/15,4/15],[0,0,1]].
#define WP4_LD 33
// RT = 3: F(4x4,3x3); RT = 4: F(3x3,4x4) — the same six points, G = rows c_k (1, p_k, .., p_k^(RT-1)), last row e_(RT-1)
template <int RT>
__device__ __forceinline__ void w4_pack_tile(const float* __restrict__ w, int64_t s_n, int64_t s_k, int64_t s_h,
                                             int64_t s_w, int flip, int N, int K, const float* __restrict__ sigma,
                                             int NT32, int Q8, float4* __restrict__ up, int qb, int nt) {
  __shared__ float g[RT][32][WP4_LD];                  // the RT row taps of one column tap b
  const int tid = threadIdx.x;                         // k range [32 qb, 32 qb + 32), n range [32 nt, 32 nt + 32)
  const bool k_fast = s_k <= s_n;
  const float sg = sigma != nullptr ? sigma[0] : 1.0f;
  const int lane = tid & 63, ql = tid >> 6;            // 4 q per block, one per wave
  const int q = qb * 4 + ql;
  const int nl = lane & 31, h = lane >> 5;
  const float Gp[5] = {0.0f, 1.0f, -1.0f, 0.5f, -2.0f};
  const float Gc[5] = {1.0f, (float)(1.0 / 3.0), (float)(-1.0 / 3.0), (float)(-16.0 / 15.0), (float)(1.0 / 15.0)};
  float G[6][RT];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float pw = 1.0f;
#pragma unroll
    for (int a = 0; a < RT; ++a) {
      G[k][a] = (float)((double)Gc[k] * (double)pw);
      pw *= Gp[k];
    }
  }
#pragma unroll
  for (int a = 0; a < RT; ++a) G[5][a] = a == RT - 1 ? 1.0f : 0.0f;
  float stage[RT * RT * 4];                             // every tap of the 32 x 32 (n, k) block: all loads in flight at once
#pragma unroll
  for (int it = 0; it < RT * RT * 4; ++it) {
    const int tap = it >> 2, r = tid + 256 * (it & 3);
    const int nl2 = k_fast ? (r >> 5) : (r & 31), kl = k_fast ? (r & 31) : (r >> 5);
    const int n = nt * 32 + nl2, k = qb * 32 + kl;
    const int a = tap / RT, b = tap - RT * a;
    const int aa = flip ? RT - 1 - a : a, bb = flip ? RT - 1 - b : b;
    stage[it] = (n < N && k < K) ? w[(int64_t)n * s_n + (int64_t)k * s_k + aa * s_h + bb * s_w] : 0.f;
  }
  float t[4][6][RT];                                    // [e][xi][b] = (G g)[xi][b]
#pragma unroll
  for (int b = 0; b < RT; ++b) {
    if (b) __syncthreads();
#pragma unroll
    for (int it = 0; it < RT * 4; ++it) {
      const int a = it >> 2, r = tid + 256 * (it & 3);
      const int nl2 = k_fast ? (r >> 5) : (r & 31), kl = k_fast ? (r & 31) : (r >> 5);
      const float v = stage[(a * RT + b) * 4 + (it & 3)];
      g[a][nl2][kl] = sigma != nullptr ? v / sg : v;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kl = 8 * ql + (e >> 1) * 4 + 2 * h + (e & 1);
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        float acc = G[xi][0] * g[0][nl][kl];
#pragma unroll
        for (int a = 1; a < RT; ++a) acc += G[xi][a] * g[a][nl][kl];
        t[e][xi][b] = acc;
      }
    }
  }
  if (q >= Q8) return;
#pragma unroll
  for (int xi = 0; xi < 6; ++xi)
#pragma unroll
    for (int nu = 0; nu < 6; ++nu) {
      float u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float acc = t[e][xi][0] * G[nu][0];
#pragma unroll
        for (int b = 1; b < RT; ++b) acc += t[e][xi][b] * G[nu][b];
        u[e] = acc;
      }
      up[(((int64_t)(xi * 6 + nu) * NT32 + nt) * Q8 + q) * 64 + lane] = make_float4(u[0], u[1], u[2], u[3]);
    }
}

template <int RT>
__global__ __launch_bounds__(256) void k_wino4_pack(const float* __restrict__ w, int64_t s_n, int64_t s_k, int64_t s_h,
                                                     int64_t s_w, int flip, int N, int K, const float* __restrict__ sigma,
                                                     int NT32, int Q8, float4* __restrict__ up) {
  w4_pack_tile<RT>(w, s_n, s_k, s_h, s_w, flip, N, K, sigma, NT32, Q8, up, blockIdx.x, blockIdx.y);
}

// several (Cout,Cin,3,3) weights per launch: block -> item by the table's block offsets (csg_pack.h)
__global__ __launch_bounds__(256) void k_wino4_pack_multi(PackMulti pm) {
  int i = 0;
  while (i + 1 < pm.n && (int)blockIdx.x >= pm.it[i + 1].start) ++i;
  const PackMultiItem& d = pm.it[i];
  const int local = blockIdx.x - d.start, nqb = (d.Q8 + 3) >> 2;
  w4_pack_tile<3>(d.w, d.s_n, d.s_k, d.s_h, d.s_w, d.flip, d.N, d.K, nullptr, d.NT32, d.Q8, d.up, local % nqb, local / nqb);
}

// Two fp32 lanes per VALU instruction (v_pk_fma_f32 / v_pk_add_f32): on this chip every VALU cycle of a SIMD is a cycle
// its matrix pipe does not get (see below), and the transforms are all pairs — (channel, channel + 1) in the input
// transform, neighbouring output channels in the output transform.  Same IEEE fma per lane as fmaf: bit-identical.
__device__ __forceinline__ csg_f32x2 w4_pfma(float c, csg_f32x2 a, csg_f32x2 b) {
  return __builtin_elementwise_fma(csg_f32x2{c, c}, a, b);
}

// ---- epilogue.  A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,0],[0,1,-1,1/8,-8,1]].
// Second half of a round: Y[a][b] = sum_xi A^T[a][xi] R_xi[b] for two output columns b, out of the exchange buffer.
// PLAIN = bias only (the SPADE gamma / beta convolutions and every backward-data pass without a gate): no per-element
// branches on the activation / residual / gate.
// MODE 0: bias only; 1: activation / residual / gate; 2: SPADE modulation (normalization.py:96-110) — the launch computes
// the BETA half of the gamma || beta convolution and writes  leaky(xhat (1 + gamma) + beta)  directly: `res` is x (the
// map being normalised, laid out like y), `gate` the gamma map an earlier launch of the same operand wrote (pixel stride
// g_cs), mod_mean / mod_invstd the batch statistics per channel.  beta never reaches memory, and the separate apply pass
// (x, gamma, beta read; y written) is gone.
// NB = output columns per round: 2 in the one-item kernel (exchange buffer over the whole staging area), 1 in the
// persistent kernel (the exchange buffer must fit in the V buffers, the raw buffers hold the next item's first stages).
template <int MODE, int NB>
__device__ __forceinline__ void w4_store_columns(const Wino4Params& p, const float* rbuf, int tig, int round, int nt32,
                                                 int img, int X0, int Y0, const float* __restrict__ bias,
                                                 const float* __restrict__ res, const float* __restrict__ gate,
                                                 float* __restrict__ y) {
  for (int item = tig; item < 256 * NB; item += W4_THREADS / 2) {  // 32 tiles x 8 channel quads x NB columns
    const int cq = item & 7, tile = (item >> 3) & 31, bb = item >> 8;
    const int n = nt32 * 32 + cq * 4;
    const int ttx = tile & (W4_TW - 1), tty = tile >> 3;
    const int oy = Y0 + 4 * tty, ox = X0 + 4 * ttx + NB * round + bb;
    if (n < p.Cout && oy < p.Ho && ox < p.Wo) {    // Ho and Wo are multiples of 4: a tile is wholly inside or outside
      csg_f32x2 lo[6], hi[6];
#pragma unroll
      for (int xi = 0; xi < 6; ++xi) {
        const csg_f32x4 r = *(const csg_f32x4*)(rbuf + ((xi * NB + bb) * 32 + tile) * W4_RSE + cq * 4);
        lo[xi] = __builtin_shufflevector(r, r, 0, 1);
        hi[xi] = __builtin_shufflevector(r, r, 2, 3);
      }
      csg_f32x2 blo = {0.f, 0.f}, bhi = {0.f, 0.f};
      if (bias != nullptr) {
        const csg_f32x4 bv = *(const csg_f32x4*)(bias + n);
        blo = __builtin_shufflevector(bv, bv, 0, 1);
        bhi = __builtin_shufflevector(bv, bv, 2, 3);
      }
      const int64_t rowstride = (int64_t)p.Wo * p.y_cs;
      const int64_t pix0 = ((int64_t)img * p.Ho + oy) * p.Wo + ox;
      int64_t off = pix0 * p.y_cs + n;
      float4 mx[4], mg[4], mm, mr;
      if (MODE == 2) {                             // the modulation's operands travel while the rows are formed
        mm = *(const float4*)(p.mod_mean + n);
        mr = *(const float4*)(p.mod_invstd + n);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          mx[a] = *(const float4*)(res + off + a * rowstride);
          mg[a] = *(const float4*)(gate + (pix0 + (int64_t)a * p.Wo) * p.g_cs + n);
        }
      }
#pragma unroll
      for (int a = 0; a < 4; ++a, off += rowstride) {
        csg_f32x2 vl, vh;
        if (a == 0) {
          vl = ((lo[0] + lo[1]) + (lo[2] + lo[3])) + lo[4];
          vh = ((hi[0] + hi[1]) + (hi[2] + hi[3])) + hi[4];
        } else if (a == 1) {
          vl = w4_pfma(-2.0f, lo[4], w4_pfma(0.5f, lo[3], lo[1] - lo[2]));
          vh = w4_pfma(-2.0f, hi[4], w4_pfma(0.5f, hi[3], hi[1] - hi[2]));
        } else if (a == 2) {
          vl = w4_pfma(4.0f, lo[4], w4_pfma(0.25f, lo[3], lo[1] + lo[2]));
          vh = w4_pfma(4.0f, hi[4], w4_pfma(0.25f, hi[3], hi[1] + hi[2]));
        } else {
          vl = w4_pfma(-8.0f, lo[4], w4_pfma(0.125f, lo[3], lo[1] - lo[2])) + lo[5];
          vh = w4_pfma(-8.0f, hi[4], w4_pfma(0.125f, hi[3], hi[1] - hi[2])) + hi[5];
        }
        vl += blo;
        vh += bhi;
        float vv[4] = {vl.x, vl.y, vh.x, vh.y};
        if (MODE == 2) {
          // as k_norm_apply_fwd: v = (x - mean) * invstd;  v = v * (1 + gamma) + beta;  LeakyReLU
          const float xs[4] = {mx[a].x, mx[a].y, mx[a].z, mx[a].w}, gs[4] = {mg[a].x, mg[a].y, mg[a].z, mg[a].w};
          const float ms[4] = {mm.x, mm.y, mm.z, mm.w}, rs[4] = {mr.x, mr.y, mr.z, mr.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (xs[e] - ms[e]) * rs[e];
            float o = xh * (1.f + gs[e]) + vv[e];
            if (p.mod_slope != 1.0f) o = o > 0.f ? o : o * p.mod_slope;
            vv[e] = o;
          }
        }
        if (MODE == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (p.act == CSG_ACT_LEAKY)
              vv[e] = vv[e] > 0.f ? vv[e] : vv[e] * p.slope;
            else if (p.act == CSG_ACT_TANH)
              vv[e] = tanhf(vv[e]);
          }
          if (res != nullptr) {
            const float4 rv = *(const float4*)(res + off);
            vv[0] += rv.x; vv[1] += rv.y; vv[2] += rv.z; vv[3] += rv.w;
          }
          if (gate != nullptr) {
            const float4 gv = *(const float4*)(gate + off);
            vv[0] *= gv.x > 0.f ? 1.f : p.gate_slope; vv[1] *= gv.y > 0.f ? 1.f : p.gate_slope;
            vv[2] *= gv.z > 0.f ? 1.f : p.gate_slope; vv[3] *= gv.w > 0.f ? 1.f : p.gate_slope;
          }
        }
        *(float4*)(y + off) = make_float4(vv[0], vv[1], vv[2], vv[3]);
      }
    }
  }
}

// MODE 3 (joint, round 6): ONE launch for a SPADE modulation.  A block's two channel groups hold the gamma tile and the beta
// tile of the SAME 32 output channels (group g reads U tile nb + g * C/32 of the ordinary gamma || beta operand), so both
// meet in the exchange buffers and  y = leaky(xhat (1 + gamma) + beta)  is formed without gamma making a round trip through
// HBM and without a second launch staging the 128-channel input again: per pixel the launch pair moved 2 Cin + 4 C floats,
// this one Cin + 3 C (gamma is still WRITTEN when `gout` is set — the backward needs it; inference passes nullptr).  A work
// item is (tile, channel quad, output column, half of the four rows): 512 * NB items over all 768 threads, each reading both
// groups' six rows.  Same expressions in the same order as MODE 0 (gamma) followed by MODE 2 (beta): bit-identical outputs.
__device__ __forceinline__ csg_f32x2 w4_out_row(int a, const csg_f32x2 (&m)[6]) {
  if (a == 0) return ((m[0] + m[1]) + (m[2] + m[3])) + m[4];
  if (a == 1) return w4_pfma(-2.0f, m[4], w4_pfma(0.5f, m[3], m[1] - m[2]));
  if (a == 2) return w4_pfma(4.0f, m[4], w4_pfma(0.25f, m[3], m[1] + m[2]));
  return w4_pfma(-8.0f, m[4], w4_pfma(0.125f, m[3], m[1] - m[2])) + m[5];
}
template <int NB>
__device__ __forceinline__ void w4_store_joint(const Wino4Params& p, const float* rbuf0, int tid, int round, int nb, int img,
                                               int X0, int Y0, const float* __restrict__ bias, const float* __restrict__ xin,
                                               float* __restrict__ gout, float* __restrict__ y) {
  const float* rbuf1 = rbuf0 + W4_RBUF / 2 * NB;            // the beta group's exchange buffer
  for (int item = tid; item < 512 * NB; item += W4_THREADS) {
    const int cq = item & 7, tile = (item >> 3) & 31, half = (item >> 8) & 1, bb = item >> 9;
    const int n = nb * 32 + cq * 4;
    const int ttx = tile & (W4_TW - 1), tty = tile >> 3;
    const int oy = Y0 + 4 * tty, ox = X0 + 4 * ttx + NB * round + bb;
    if (oy < p.Ho && ox < p.Wo) {                            // Ho and Wo are multiples of 4; n < C by construction
      // gamma first, then beta (one group's six rows live at a time: the one-item form still carries its accumulators here)
      csg_f32x2 gl2[2], gh2[2], bl2[2], bh2[2];
      {
        csg_f32x2 lo[6], hi[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const csg_f32x4 g = *(const csg_f32x4*)(rbuf0 + ((xi * NB + bb) * 32 + tile) * W4_RSE + cq * 4);
          lo[xi] = __builtin_shufflevector(g, g, 0, 1);
          hi[xi] = __builtin_shufflevector(g, g, 2, 3);
        }
        const csg_f32x4 bg = *(const csg_f32x4*)(bias + n);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          gl2[r] = w4_out_row(2 * half + r, lo) + csg_f32x2{bg.x, bg.y};
          gh2[r] = w4_out_row(2 * half + r, hi) + csg_f32x2{bg.z, bg.w};
        }
      }
      {
        csg_f32x2 lo[6], hi[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const csg_f32x4 b = *(const csg_f32x4*)(rbuf1 + ((xi * NB + bb) * 32 + tile) * W4_RSE + cq * 4);
          lo[xi] = __builtin_shufflevector(b, b, 0, 1);
          hi[xi] = __builtin_shufflevector(b, b, 2, 3);
        }
        const csg_f32x4 bbv = *(const csg_f32x4*)(bias + p.Cout + n);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          bl2[r] = w4_out_row(2 * half + r, lo) + csg_f32x2{bbv.x, bbv.y};
          bh2[r] = w4_out_row(2 * half + r, hi) + csg_f32x2{bbv.z, bbv.w};
        }
      }
      const float4 mm = *(const float4*)(p.mod_mean + n), mr = *(const float4*)(p.mod_invstd + n);
      const int64_t rowstride = (int64_t)p.Wo * p.y_cs;
      const int64_t pix0 = ((int64_t)img * p.Ho + oy + 2 * half) * p.Wo + ox;
      const int64_t off = pix0 * p.y_cs + n;
      float4 mx[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) mx[r] = *(const float4*)(xin + off + r * rowstride);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const csg_f32x2 gl = gl2[r], gh = gh2[r], bl = bl2[r], bh = bh2[r];
        const float gs[4] = {gl.x, gl.y, gh.x, gh.y}, vv[4] = {bl.x, bl.y, bh.x, bh.y};
        const float xs[4] = {mx[r].x, mx[r].y, mx[r].z, mx[r].w};
        const float ms[4] = {mm.x, mm.y, mm.z, mm.w}, rs[4] = {mr.x, mr.y, mr.z, mr.w};
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xh = (xs[e] - ms[e]) * rs[e];
          float o = xh * (1.f + gs[e]) + vv[e];
          if (p.mod_slope != 1.0f) o = o > 0.f ? o : o * p.mod_slope;
          out[e] = o;
        }
        if (gout != nullptr)
          *(float4*)(gout + (pix0 + (int64_t)r * p.Wo) * p.g_cs + n) = make_float4(gs[0], gs[1], gs[2], gs[3]);
        *(float4*)(y + off + r * rowstride) = make_float4(out[0], out[1], out[2], out[3]);
      }
    }
  }
}

template <int NB>
__device__ __forceinline__ void w4_epilogue(f32x16 (&acc)[6], const Wino4Params& p, float* rbase, int tid, int wave, int grp,
                                            int j, int h, int nt32, int img, int X0, int Y0, const float* __restrict__ bias,
                                            const float* __restrict__ res, const float* __restrict__ gate,
                                            float* __restrict__ y) {
  // Per wave (row xi): R[b] = sum_nu M[xi][nu] A^T[b][nu]; then Y[a][b] = sum_xi A^T[a][xi] R_xi[b] through LDS, NB
  // output columns b per round: rbuf[xi][b % NB][32 tiles][W4_RSE]
  float* rbuf = rbase + grp * (W4_RBUF / 2 * NB);   // this channel group's exchange buffer
  const int tig = tid - grp * (W4_THREADS / 2);  // thread index inside the channel group
  const bool plain = p.act == CSG_ACT_NONE && res == nullptr && gate == nullptr;
  if (NB == 2) {
    __syncthreads();                             // every wave is done reading the staging buffers
    W4_T(3)
  }
  // NB = 1: all four columns are reduced over nu FIRST (64 registers in place of the 96 accumulators), so that the three
  // store phases that follow a not-yet-written column do not have to carry the accumulators next to their own operands
  csg_f32x2 rr[NB == 1 ? 4 : 1][8];
  if (NB == 1) {
#pragma unroll
    for (int ge = 0; ge < 8; ++ge) {
#define W4_M(K) csg_f32x2{acc[K][2 * ge], acc[K][2 * ge + 1]}
      const csg_f32x2 m0 = W4_M(0), m1 = W4_M(1), m2 = W4_M(2), m3 = W4_M(3), m4 = W4_M(4), m5 = W4_M(5);
#undef W4_M
      rr[0][ge] = ((m0 + m1) + (m2 + m3)) + m4;
      rr[NB == 1 ? 1 : 0][ge] = w4_pfma(-2.0f, m4, w4_pfma(0.5f, m3, m1 - m2));
      rr[NB == 1 ? 2 : 0][ge] = w4_pfma(4.0f, m4, w4_pfma(0.25f, m3, m1 + m2));
      rr[NB == 1 ? 3 : 0][ge] = w4_pfma(-8.0f, m4, w4_pfma(0.125f, m3, m1 - m2)) + m5;
    }
    // pinned here: the compiler would otherwise sink the later columns into their rounds and keep the accumulators alive
#pragma unroll
    for (int b = 0; b < (NB == 1 ? 4 : 1); ++b)
#pragma unroll
      for (int ge = 0; ge < 8; ++ge) asm volatile("" : "+v"(rr[b][ge]));
  }
#pragma unroll
  for (int round = 0; round < 4 / NB; ++round) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      csg_f32x2 r0[2], r1[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (NB == 1) {
          r0[e] = rr[NB == 1 ? round : 0][2 * g + e];
          continue;
        }
#define W4_M(K) csg_f32x2{acc[K][4 * g + 2 * e], acc[K][4 * g + 2 * e + 1]}
        const csg_f32x2 m0 = W4_M(0), m1 = W4_M(1), m2 = W4_M(2), m3 = W4_M(3), m4 = W4_M(4), m5 = W4_M(5);
#undef W4_M
        const int b0 = NB * round;               // first output column of this round
        if (b0 == 0) r0[e] = ((m0 + m1) + (m2 + m3)) + m4;
        if (b0 == 2) r0[e] = w4_pfma(4.0f, m4, w4_pfma(0.25f, m3, m1 + m2));
        if (b0 == 0) r1[e] = w4_pfma(-2.0f, m4, w4_pfma(0.5f, m3, m1 - m2));
        if (b0 == 2) r1[e] = w4_pfma(-8.0f, m4, w4_pfma(0.125f, m3, m1 - m2)) + m5;
      }
      const int ch = 8 * g + 4 * h;
      *(float4*)(rbuf + ((wave * NB + 0) * 32 + j) * W4_RSE + ch) = make_float4(r0[0].x, r0[0].y, r0[1].x, r0[1].y);
      if (NB == 2)
        *(float4*)(rbuf + ((wave * NB + 1) * 32 + j) * W4_RSE + ch) = make_float4(r1[0].x, r1[0].y, r1[1].x, r1[1].y);
    }
    __syncthreads();
    if (NB == 2 && round == 0) { W4_T(4) }
    if (NB == 1 && p.mod == 2)            // (the persistent form only: the one-item form carries its accumulators through the rounds)
      w4_store_joint<NB>(p, rbase, tid, round, nt32 - grp * p.gb_off, img, X0, Y0, bias, res, const_cast<float*>(gate), y);
    else if (p.mod)
      w4_store_columns<2, NB>(p, rbuf, tig, round, nt32, img, X0, Y0, bias, res, gate, y);
    else if (plain)
      w4_store_columns<0, NB>(p, rbuf, tig, round, nt32, img, X0, Y0, bias, res, gate, y);
    else
      w4_store_columns<1, NB>(p, rbuf, tig, round, nt32, img, X0, Y0, bias, res, gate, y);
    if (round + 1 < 4 / NB) {
      __syncthreads();                           // the exchange buffer is rewritten by the next round
      if (NB == 2) { W4_T(5) }
    }
  }
  if (NB == 2) { W4_T(6) }
}

// ---- epilogue of F(3x3,4x4): A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,1]] (the first three rows of the
// F(4x4,3x3) matrix, the point at infinity moved into the last one).  Round 0 forms output columns 0 and 1 of every
// 3 x 3 tile, round 1 column 2; output sizes need not be multiples of 3: every pixel is bounds-checked.
__device__ __forceinline__ void w3_epilogue(f32x16 (&acc)[6], const Wino4Params& p, float* smem, int tid, int wave, int grp,
                                            int j, int h, int nt32, int img, int X0, int Y0, const float* __restrict__ bias,
                                            const float* __restrict__ res, const float* __restrict__ gate,
                                            float* __restrict__ y) {
  float* rbuf = smem + grp * W4_RBUF;
  const int tig = tid - grp * (W4_THREADS / 2);
  __syncthreads();                               // every wave is done reading the staging buffers
  W4_T(3)
#pragma unroll
  for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      csg_f32x2 r0[2], r1[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
#define W4_M(K) csg_f32x2{acc[K][4 * g + 2 * e], acc[K][4 * g + 2 * e + 1]}
        const csg_f32x2 m0 = W4_M(0), m1 = W4_M(1), m2 = W4_M(2), m3 = W4_M(3), m4 = W4_M(4), m5 = W4_M(5);
#undef W4_M
        if (round == 0) {
          r0[e] = ((m0 + m1) + (m2 + m3)) + m4;
          r1[e] = w4_pfma(-2.0f, m4, w4_pfma(0.5f, m3, m1 - m2));
        } else {
          r0[e] = w4_pfma(4.0f, m4, w4_pfma(0.25f, m3, m1 + m2)) + m5;
        }
      }
      const int ch = 8 * g + 4 * h;
      *(float4*)(rbuf + ((wave * 2 + 0) * 32 + j) * W4_RSE + ch) = make_float4(r0[0].x, r0[0].y, r0[1].x, r0[1].y);
      if (round == 0)
        *(float4*)(rbuf + ((wave * 2 + 1) * 32 + j) * W4_RSE + ch) = make_float4(r1[0].x, r1[0].y, r1[1].x, r1[1].y);
    }
    __syncthreads();
    if (round == 0) { W4_T(4) }
    const int nitem = round == 0 ? 512 : 256;      // 32 tiles x 8 channel quads x (2 | 1) columns
    for (int item = tig; item < nitem; item += W4_THREADS / 2) {
      const int cq = item & 7, tile = (item >> 3) & 31, bb = item >> 8;
      const int n = nt32 * 32 + cq * 4;
      const int ttx = tile & (W4_TW - 1), tty = tile >> 3;
      const int oy = Y0 + 3 * tty, ox = X0 + 3 * ttx + 2 * round + bb;
      if (n < p.Cout && oy < p.Ho && ox < p.Wo) {
        csg_f32x2 lo[6], hi[6];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
          const csg_f32x4 r = *(const csg_f32x4*)(rbuf + ((xi * 2 + bb) * 32 + tile) * W4_RSE + cq * 4);
          lo[xi] = __builtin_shufflevector(r, r, 0, 1);
          hi[xi] = __builtin_shufflevector(r, r, 2, 3);
        }
        csg_f32x2 blo = {0.f, 0.f}, bhi = {0.f, 0.f};
        if (bias != nullptr) {
          const csg_f32x4 bv = *(const csg_f32x4*)(bias + n);
          blo = __builtin_shufflevector(bv, bv, 0, 1);
          bhi = __builtin_shufflevector(bv, bv, 2, 3);
        }
        const int64_t rowstride = (int64_t)p.Wo * p.y_cs;
        int64_t off = (((int64_t)img * p.Ho + oy) * p.Wo + ox) * p.y_cs + n;
#pragma unroll
        for (int a = 0; a < 3; ++a, off += rowstride) {
          if (oy + a >= p.Ho) break;
          csg_f32x2 vl, vh;
          if (a == 0) {
            vl = ((lo[0] + lo[1]) + (lo[2] + lo[3])) + lo[4];
            vh = ((hi[0] + hi[1]) + (hi[2] + hi[3])) + hi[4];
          } else if (a == 1) {
            vl = w4_pfma(-2.0f, lo[4], w4_pfma(0.5f, lo[3], lo[1] - lo[2]));
            vh = w4_pfma(-2.0f, hi[4], w4_pfma(0.5f, hi[3], hi[1] - hi[2]));
          } else {
            vl = w4_pfma(4.0f, lo[4], w4_pfma(0.25f, lo[3], lo[1] + lo[2])) + lo[5];
            vh = w4_pfma(4.0f, hi[4], w4_pfma(0.25f, hi[3], hi[1] + hi[2])) + hi[5];
          }
          vl += blo;
          vh += bhi;
          float vv[4] = {vl.x, vl.y, vh.x, vh.y};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if (p.act == CSG_ACT_LEAKY)
              vv[e] = vv[e] > 0.f ? vv[e] : vv[e] * p.slope;
            else if (p.act == CSG_ACT_TANH)
              vv[e] = tanhf(vv[e]);
          }
          if (res != nullptr) {
            const float4 rv = *(const float4*)(res + off);
            vv[0] += rv.x; vv[1] += rv.y; vv[2] += rv.z; vv[3] += rv.w;
          }
          if (gate != nullptr) {
            const float4 gv = *(const float4*)(gate + off);
            vv[0] *= gv.x > 0.f ? 1.f : p.gate_slope; vv[1] *= gv.y > 0.f ? 1.f : p.gate_slope;
            vv[2] *= gv.z > 0.f ? 1.f : p.gate_slope; vv[3] *= gv.w > 0.f ? 1.f : p.gate_slope;
          }
          *(float4*)(y + off) = make_float4(vv[0], vv[1], vv[2], vv[3]);
        }
      }
    }
    if (round == 0) {
      __syncthreads();                           // the exchange buffer is rewritten by round 1
      W4_T(5)
    }
  }
  W4_T(6)
}

// ------------------------------------------------------------------------------------ convolution
// On this chip the fp32 MFMA and the fp32 VALU do not overlap on a SIMD the way the bf16 matrix pipe and the VALU do:
// the ablations of k_wino4_conv add up (transform without MFMAs 0.25 ms + MFMAs with the column transform 0.29 ms =
// 0.48 ms measured, software-pipelined or not), i.e. every VALU instruction of the transform costs matrix-pipe time.
// So the transform must be computed ONCE: in k_wino4_conv both channel groups of a block form the same V in registers
// (~8 VALU per MFMA).  Here the twelve waves split the transform of a stage without redundancy — wave w forms row
// xi = w mod 6 of V for the channel pairs 2 (w / 6) + {0, 1} (one per half-wave) — and write it to LDS in MFMA operand
// order; a stage later every wave reads the six operands of its own row with one conflict-free ds_read_b128 each
// (four MFMAs per read).  ~4 VALU per MFMA, and the transform of stage s+1 is independent of the MFMAs of stage s.
//   LDS (words): raw[2] (2 x W4_BUFW) | V[2] (2 x 9216: [xi][nu][half h][tile][4]) | per-thread offset table.
//   Stage k (local index): produce V[k+1] from raw[k+1]; MFMAs out of V[k]; store raw[k+2]; barrier; load raw[k+3].
#define W4_VBUF (36 * 2 * 32 * 4)
// Geometry of the staged region for output tiles of T x T pixels out of 6 x 6 input tiles: T = 4 is F(4x4,3x3), T = 3 is
// F(3x3,4x4) (the PatchGAN's 4x4 / stride 1 layers) — same points, same input transform, same 36 positions.  Columns
// are de-interleaved modulo T and the row groups of T rows skewed by two words: the words of the 32 tiles a half-wave
// reads with one ds_read_b64 sit in 32 different bank pairs (T * RS is a multiple of 64 words).
template <int T>
struct W4Geo {
  static constexpr int R = T * W4_TH + 6 - T;          // staged rows: 18 / 15
  static constexpr int C = T * W4_TW + 6 - T;          // staged columns: 34 / 27
  static constexpr int RS = T == 4 ? 288 : 256;        // words per staged row (T classes x 9 columns x 8 words, padded)
  static constexpr int BUFW = R * RS + 8 + 16;         // + the skew of the last row group + a dump slot
  static constexpr int V0 = 2 * BUFW;
  static constexpr int OFFTAB = (2 * BUFW + 2 * W4_VBUF) > 2 * W4_RBUF ? (2 * BUFW + 2 * W4_VBUF) : 2 * W4_RBUF;
  static_assert((T * RS) % 64 == 0 && R * C * 2 <= W4_NLD * W4_THREADS && (C + T - 1) / T <= W4_CQ && BUFW < 65536, "staging geometry");
};
#define W4V_RAW0 0
// the kernel-argument segment of k_wino4_conv_v as the persistent form reads it back in its epilogue
struct W4KernArgs {
  Wino4Params p;
  const float* x;
  const float4* up;
  const float* bias;
  const float* res;
  const float* gate;
  float* y;
};
typedef const __attribute__((address_space(4))) W4KernArgs* w4_kernargs_ptr;
// P = persistent: 256 blocks (one per CU) walk the (region, channel block) items of their XCD's share, and the stream of
// stages runs ACROSS items — the last two stages of an item store the first two stages of the next one (raw[0], raw[1]),
// the third is in flight and the U ring already holds the next item's operands when the epilogue starts; the epilogue
// exchanges one output column per round inside the V buffers.  What a one-item block pays outside its loop on a
// Cin = 128 layer (profiles/archive/r05p_conv_phase_trace.txt: 7.7k cycles until the first stage lands + 2.6k to prime, of
// 121k) shrinks to the transform of the first stage.  Same arithmetic in the same order: bit-identical outputs.
template <int T, bool P>
__global__ __launch_bounds__(W4_THREADS, 3) void k_wino4_conv_v(Wino4Params p, const float* __restrict__ x,
                                                                 const float4* __restrict__ up,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ res,
                                                                 const float* __restrict__ gate, float* __restrict__ y) {
  typedef W4Geo<T> Geo;
  constexpr int W4_RS = Geo::RS, W4_BUFW = Geo::BUFW, W4V_V0 = Geo::V0, W4V_OFFTAB = Geo::OFFTAB;
  static_assert(!P || T == 4, "the persistent form serves F(4x4,3x3)");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  W4_T(0)
  const int wave12 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave12 >= 6 ? 1 : 0;           // consumer: channel group; producer: channel pairs 2 grp + {0, 1}
  const int wave = wave12 - 6 * grp;             // row xi of the transformed domain (both roles)

  // items: (image, region row, region column, channel block[, input-channel range]); the channel block runs fastest, so
  // the blocks of an XCD that run together read the same input region out of its L2
  int nb, img, X0, Y0, split = 0;
  auto decode = [&](int item) {
    if (!P) {
      split = item % p.ksplit;
      item /= p.ksplit;
    }
    nb = item % p.nblocks;
    item /= p.nblocks;
    const int bx = item % p.tbx;
    item /= p.tbx;
    const int by = item % p.tby;
    img = item / p.tby;
    X0 = bx * T * W4_TW;
    Y0 = by * T * W4_TH;
  };
  // persistent walk: XCD x (blockIdx & 7) owns the contiguous share [v_lo, v_hi) of the items, block idx of that XCD
  // takes every (gridDim / 8)-th of them — at any time the 32 CUs of an XCD work on 32 consecutive items
  int v = 0, v_hi = 0, v_step = 1;
  if (P) {
    const int q = p.nitems >> 3, r = p.nitems & 7, xcd = blockIdx.x & 7;
    const int v_lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    v_hi = v_lo + q + (xcd < r ? 1 : 0);
    v_step = gridDim.x >> 3;
    v = v_lo + (blockIdx.x >> 3);
    if (v >= v_hi) return;
    decode(v);
  } else {
    decode(w4_xcd_remap(blockIdx.x, gridDim.x));
  }

  const csg_i32x4 rsX = csg_make_srd(x, (long long)p.B * p.H * p.W * p.x_cs * 4);
  const csg_i32x4 rsU = csg_make_srd(up, (long long)36 * p.NT32 * p.Q8 * 64 * 16);

  // ---- staging plan (as k_wino4_conv: two 16-byte pieces per thread and stage, offsets parked in LDS); the persistent
  // kernel keeps two tables, this item's and the next one's (every thread reads its own entry only: no barrier)
  unsigned* const s_tab = (unsigned*)(smem + W4V_OFFTAB) + tid * 4;
  int tsel = 0;                                  // which of the two tables is this item's
#define W4_TAB(k) (s_tab + (P ? (k) * (W4_THREADS * 4) : 0))
  auto fill_plan = [&](unsigned* tab, bool valid) {
    unsigned goff[W4_NLD], loffp = 0;
    // persistent: the thread index is recovered from the table address (one register less across the main loop) and made
    // opaque, so that nothing of the plan is hoisted out of the item loop and kept
    int tid_o = P ? (int)(s_tab - (unsigned*)(smem + W4V_OFFTAB)) >> 2 : tid;
    if (P) asm volatile("" : "+v"(tid_o));
#pragma unroll
    for (int i = 0; i < W4_NLD; ++i) {
      const int e = tid_o + W4_THREADS * i;
      goff[i] = CSG_OOB_OFF;
      int lo = (W4_BUFW - 16) + (tid_o & 3) * 4;     // the dump slot
      if (e < Geo::R * Geo::C * 2) {
        const int pix = e >> 1, c4 = e & 1;
        const int row = pix / Geo::C, col = pix - row * Geo::C;
        const int iy = Y0 + row - p.pad, ix = X0 + col - p.pad;
        // the float index of the piece inside a raw buffer, skew of its row group included (16 bits: BUFW < 65536) — the
        // per-stage store decodes nothing (round 6: +1-2 % on every shape over a packed quad index + skew code)
        lo = row * W4_RS + ((col % T) * W4_CQ + (col / T)) * W4_PS + c4 * 4 + (row / T) * 2;
        if (valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
          goff[i] = (unsigned)(((img * p.H + iy) * p.W + ix) * p.x_cs + c4 * 4) * 4u;
      }
      loffp |= (unsigned)lo << (16 * i);
    }
    *(uint4*)tab = make_uint4(goff[0], goff[1], loffp, 0u);
  };
  fill_plan(W4_TAB(0), true);
  const int s_begin = split * p.sps, s_end = min(p.nstage, s_begin + p.sps);
  csg_f32x4 st[W4_NLD];
  // one-item kernel: s past the end loads finite garbage or zeros, never consumed; persistent: the next item's stages
  // (only stages 0 and 1: they are stored by this item's last two stages; stage 2 is fetched behind the epilogue)
  // Every call issues its two loads (the last stage of an item: stage 2 of the next one, carried across the epilogue):
  // the wait counts in front of the MFMAs are only exact when the number of loads in flight is the same on every path.
  auto load_stage = [&](int s) {
    const bool over = P && s >= s_end;
    const uint2 go = *(const uint2*)W4_TAB(over ? tsel ^ 1 : tsel);
    if (over) s -= s_end;
    st[0] = csg_buf_load_x4(rsX, (int)go.x, s * (W4_PS * 4), 0);
    st[1] = csg_buf_load_x4(rsX, (int)go.y, s * (W4_PS * 4), 0);
  };
  auto store_stage = [&](float* base) {
    const unsigned lp = s_tab[2];
#pragma unroll
    for (int i = 0; i < W4_NLD; ++i) {
      const unsigned lo = (i & 1) ? (lp >> 16) : (lp & 0xffffu);
      float* dst = base + lo;
      *(float2*)dst = make_float2(st[i].x, st[i].y);
      *(float2*)(dst + 2) = make_float2(st[i].z, st[i].w);
    }
  };

  const int j = lane & 31, h = lane >> 5;
  const int tx = j & (W4_TW - 1), ty = j >> 3;
  const int rb = wave == 0 ? 0 : 1;
  // producer: raw words of (row T ty + rb, column T tx, channel pair q = 2 grp + h); rows rb + i of the window sit in the
  // next row group (two more words of skew) from i = T - rb on
  // (the offsets of the later rows are compile-time constants inside the per-xi copies of the transform: one base register)
  const float* p0 = smem + W4V_RAW0 + (T * ty + rb) * W4_RS + 2 * ty + tx * W4_PS + 4 * grp + 2 * h;
  // producer: V words of (xi = wave, nu = 0, half h, tile j), channel pair slot grp;  consumer: the same row, float4
  float* pv = smem + W4V_V0 + (((wave * 6) * 2 + h) * 32 + j) * 4;

  // U tile of this wave's channel group: neighbours, or (joint gamma | beta launch) the gamma tile nb and its beta tile
  int nt32 = p.gb_off ? nb + grp * p.gb_off : nb * 2 + grp;
  // U operand of (position (xi = wave, nu), k-oct q, this wave's 32-channel tile): wave-uniform base in a scalar
  // register, the lane's 16 bytes in a vector register (out of range when the tile lies beyond Cout)
  auto u_tile = [&]() { return ((wave * 6) * p.NT32 + nt32 + p.nt_off) * p.Q8 * 1024; };
  const unsigned ulane = nt32 + p.nt_off < p.NT32 || P ? (unsigned)lane * 16u : CSG_OOB_OFF;   // P: Cout % 64 == 0
  int ubase = u_tile(), ubase_nxt = ubase;
  const int ustride = p.NT32 * p.Q8 * 1024;
  csg_f32x4 ur[3];                               // ring: the operands of position nu are fetched three positions ahead
  auto load_ur = [&](int slot, int nu, int s) {  // persistent, s past the end: the first k-oct of the next item
    const bool over = P && s >= s_end;
    const int qq = over ? 0 : min(s, p.Q8 - 1);
    ur[slot] = csg_buf_load_x4(rsU, (int)ulane, (over ? ubase_nxt : ubase) + nu * ustride + qq * 1024, 0);
  };

  f32x16 acc[6];

  // V[xi = wave][0..5] of (tile j, channel pair 2 grp + h) out of raw buffer rbufsel, into V buffer vbufsel.  The row
  // combination is specialised per xi (a scalar switch around six copies of this code): rows 1..4 of B^T touch four
  // input rows and every row factors into 3-4 operations where the generic five-coefficient chain costs 5.
  //   xi = 0, 5: (e0 + e4) - 2 e2 + 1.5 (e3 - e1)      (window e = d0..d4 / d1..d5)
  //   xi = 1: (e3 - e0) + 0.5 e1 + 2.5 e2    xi = 2: (e3 + e0) - 2.5 e1 + 0.5 e2      (window e = d1..d4)
  //   xi = 3: (e3 - e1) + 2 (e2 - e0)        xi = 4: (e3 - e1) - 0.5 (e2 - e0)
  auto produce_xi = [&](auto xi_tag, int rbufsel, int vbufsel) {
    constexpr int XI = decltype(xi_tag)::value;
    constexpr int RBC = XI == 0 ? 0 : 1;         // = rb of this wave
    constexpr int O2 = 2 * W4_RS + (T == 3 ? 2 * RBC : 0), O3 = 3 * W4_RS + (T == 3 ? 2 : 2 * RBC);
    csg_f32x2 t[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const int co = rbufsel * W4_BUFW + ((c % T) * W4_CQ + (c / T)) * W4_PS;
      const csg_f32x2 e0 = *(w4_lds_cv2)(p0 + co);
      const csg_f32x2 e1 = *(w4_lds_cv2)(p0 + co + W4_RS);
#ifdef W4_HALF_READS   // ablation (wrong results): half the raw LDS reads of the transform at the same VALU count
      const csg_f32x2 e2 = e0 * 1.5f;
      const csg_f32x2 e3 = e1 * 0.75f;
#else
      const csg_f32x2 e2 = *(w4_lds_cv2)(p0 + co + O2);
      const csg_f32x2 e3 = *(w4_lds_cv2)(p0 + co + O3);
#endif
      if (XI == 0 || XI == 5) {
#ifdef W4_HALF_READS
        const csg_f32x2 e4 = e0 * 0.5f;
#else
        const csg_f32x2 e4 = *(w4_lds_cv2)(p0 + co + 4 * W4_RS + 2);
#endif
        t[c] = w4_pfma(1.5f, e3 - e1, w4_pfma(-2.0f, e2, e0 + e4));
      } else if (XI == 1) {
        t[c] = w4_pfma(2.5f, e2, w4_pfma(0.5f, e1, e3 - e0));
      } else if (XI == 2) {
        t[c] = w4_pfma(0.5f, e2, w4_pfma(-2.5f, e1, e3 + e0));
      } else if (XI == 3) {
        t[c] = w4_pfma(2.0f, e2 - e0, e3 - e1);
      } else {
        t[c] = w4_pfma(-0.5f, e2 - e0, e3 - e1);
      }
    }
    csg_f32x2 v[6];
    {
      const csg_f32x2 s42 = t[4] - t[2], s31 = t[3] - t[1];
      v[0] = w4_pfma(1.5f, s31, w4_pfma(-2.0f, t[2], t[0] + t[4]));
      v[1] = w4_pfma(2.5f, t[3], w4_pfma(0.5f, t[2], t[4] - t[1]));
      v[2] = w4_pfma(0.5f, t[3], w4_pfma(-2.5f, t[2], t[4] + t[1]));
      v[3] = w4_pfma(2.0f, s31, s42);
      v[4] = w4_pfma(-0.5f, s31, s42);
      v[5] = w4_pfma(1.5f, s42, w4_pfma(-2.0f, t[3], t[1] + t[5]));
    }
    // persistent: the write address is formed here from the read base (an opaque zero keeps it from being hoisted into a
    // register of its own that lives across the whole item loop — the kernel sits at its register limit)
    int goff2 = 2 * grp;
    if (P) {
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      goff2 += z;
    }
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
      *(csg_f32x2*)(pv + vbufsel * W4_VBUF + nu * 256 + goff2) = v[nu];
  };
  auto produce = [&](int rbufsel, int vbufsel) {
    switch (wave) {
      case 0: produce_xi(std::integral_constant<int, 0>(), rbufsel, vbufsel); break;
      case 1: produce_xi(std::integral_constant<int, 1>(), rbufsel, vbufsel); break;
      case 2: produce_xi(std::integral_constant<int, 2>(), rbufsel, vbufsel); break;
      case 3: produce_xi(std::integral_constant<int, 3>(), rbufsel, vbufsel); break;
      case 4: produce_xi(std::integral_constant<int, 4>(), rbufsel, vbufsel); break;
      default: produce_xi(std::integral_constant<int, 5>(), rbufsel, vbufsel); break;
    }
  };
  // MFMAs of positions nu in [LO, HI) out of V buffer vbufsel with the U operands of k-oct s (four per position);
  // refills the ring for k-oct s (positions 3..5) and s + 1 (positions 0..2)
  auto consume = [&](auto lo_tag, auto hi_tag, int vbufsel, int s) {
#pragma unroll
    for (int nu = decltype(lo_tag)::value; nu < decltype(hi_tag)::value; ++nu) {
      const float4 v = *(const float4*)(pv + vbufsel * W4_VBUF + nu * 256);
      const int slot = nu % 3;
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[slot].x, v.x, acc[nu], 0, 0, 0);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[slot].y, v.y, acc[nu], 0, 0, 0);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[slot].z, v.z, acc[nu], 0, 0, 0);
      acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur[slot].w, v.w, acc[nu], 0, 0, 0);
#ifndef W4_NO_ULOAD
      if (nu < 3) load_ur(slot, nu + 3, s); else load_ur(slot, nu - 3, s + 1);
#endif
    }
  };
  // One stage.  The transform of stage k+1 sits BETWEEN the MFMAs of stage k (W4_SPLIT positions in front of it): after
  // the barrier every wave has matrix work at once (V[k] is complete) and again after its transform, instead of all
  // twelve waves issuing their 166 KB of transform reads together while the matrix pipes wait.
#ifndef W4_SPLIT
#define W4_SPLIT 3
#endif
  // (-DW4_HALF_PRODUCE, -DW4_HALF_READS, -DW4_NO_BARRIER, -DW4_NO_ULOAD, -DW4_NO_STAGING: ablations for
  // tools/wino4_variants.py — wrong results, loop timing only.
  // Round 6 measured two more placements with that tool and dropped them: s_setprio 1 / 3 around the matrix clusters
  // -3 % on every shape; different split points on the three waves of a SIMD needed the stage body as a callable the
  // compiler no longer inlined — 1 KB of scratch, 30x slower.  profiles/r06b_wino4_variants.txt.)
  auto stage = [&](int s, auto par_tag) {        // par = (s - s_begin) & 1
    constexpr int par = decltype(par_tag)::value;
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, W4_SPLIT> IS;
    typedef std::integral_constant<int, 6> I6;
#ifndef W4_NO_MFMA
    consume(I0(), IS(), par, s);
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef W4_NO_PRODUCE
#ifdef W4_HALF_PRODUCE
    if (s + 1 < s_end && par == 0) produce(par ^ 1, par ^ 1);
#else
    if (s + 1 < s_end) produce(par ^ 1, par ^ 1);   // V[k+1] from raw[k+1]
#endif
#endif
    __builtin_amdgcn_sched_barrier(0);
#ifndef W4_NO_MFMA
    consume(IS(), I6(), par, s);
#endif
#if !defined(W4_NO_STAGING) && !defined(W4_NO_STAGING_STORE)
    store_stage(smem + W4V_RAW0 + par * W4_BUFW);   // raw[k+2] takes the buffer raw[k] left
#endif
#ifndef W4_NO_BARRIER
    __syncthreads();
#endif
#if !defined(W4_NO_STAGING) && !defined(W4_NO_STAGING_LOAD)
    load_stage(s + 3);
#endif
  };

  y += (long long)split * p.slab;
  {
    // prologue: the first two stages travel together (one exposed memory latency instead of two)
    const uint2 go = *(const uint2*)W4_TAB(0);
    const csg_f32x4 a0 = csg_buf_load_x4(rsX, (int)go.x, s_begin * (W4_PS * 4), 0);
    const csg_f32x4 a1 = csg_buf_load_x4(rsX, (int)go.y, s_begin * (W4_PS * 4), 0);
    load_stage(s_begin + 1);
#pragma unroll
    for (int nu = 0; nu < 3; ++nu) load_ur(nu, nu, s_begin);
    const csg_f32x4 b0 = st[0], b1 = st[1];
    st[0] = a0;
    st[1] = a1;
    store_stage(smem + W4V_RAW0);
    __syncthreads();
    W4_T(1)
    st[0] = b0;
    st[1] = b1;
    produce(0, 0);                               // V[0] from raw[0]
    store_stage(smem + W4V_RAW0 + W4_BUFW);      // raw[1]
    __syncthreads();
    load_stage(s_begin + 2);
  }
  W4_T(2)
  for (;;) {
#pragma unroll
    for (int nu = 0; nu < 6; ++nu)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nu][e] = 0.f;
    bool more = false;
    int item_n = 0;
    W4_TI(0)
    int nb_n = nb, img_n = img, X0_n = X0, Y0_n = Y0;
    if (P) {
      // the next item's plan: its first two stages enter the pipeline three stages before this item ends
      const int nb_c = nb, img_c = img, X0_c = X0, Y0_c = Y0;
      more = v + v_step < v_hi;
      item_n = more ? v + v_step : v;
      if (more) decode(item_n);
      fill_plan(W4_TAB(tsel ^ 1), more);
      nt32 = p.gb_off ? nb + grp * p.gb_off : nb * 2 + grp;
      ubase_nxt = u_tile();
      nb_n = nb; img_n = img; X0_n = X0; Y0_n = Y0;
      nb = nb_c; img = img_c; X0 = X0_c; Y0 = Y0_c;
      nt32 = p.gb_off ? nb + grp * p.gb_off : nb * 2 + grp;
    }
    int s = s_begin;
    W4_TI(1)
    for (; s + 1 < s_end; s += 2) {
      stage(s, std::integral_constant<int, 0>());
      stage(s + 1, std::integral_constant<int, 1>());
#ifdef W4_TRACE
      if (s == s_begin) { W4_TI(2) }
      if (s + 4 == s_end) { W4_TI(4) }
#endif
    }
    if (s < s_end) stage(s, std::integral_constant<int, 0>());
    W4_TI(3)
    if (T == 4 && P) {
      // the epilogue's addresses and parameters are formed HERE, not kept across the main loop: the thread index goes
      // through an opaque copy and the parameters are read back from the kernel-argument segment (scalar loads)
      int tid_o = (int)(s_tab - (unsigned*)(smem + W4V_OFFTAB)) >> 2;
      asm volatile("" : "+v"(tid_o));
      w4_kernargs_ptr ka = (w4_kernargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      Wino4Params pe = p;
#define W4_F(f) pe.f = ka->p.f;
      W4_F(Cout) W4_F(y_cs) W4_F(Ho) W4_F(Wo) W4_F(act) W4_F(slope) W4_F(gate_slope) W4_F(mod) W4_F(g_cs) W4_F(gb_off) W4_F(mod_slope)
      W4_F(mod_mean) W4_F(mod_invstd)
#undef W4_F
      w4_epilogue<1>(acc, pe, smem + W4V_V0, tid_o, wave, grp, tid_o & 31, (tid_o >> 5) & 1, nt32, img, X0, Y0, ka->bias,
                     ka->res, ka->gate, ka->y);
    } else if (T == 4) {
      w4_epilogue<2>(acc, p, smem, tid, wave, grp, j, h, nt32, img, X0, Y0, bias, res, gate, y);
    } else {
      w3_epilogue(acc, p, smem, tid, wave, grp, j, h, nt32, img, X0, Y0, bias, res, gate, y);
    }
    W4_TI(6)
    if (!P || !more) break;
    // the next item: raw[0] and raw[1] are in place, its U ring is loaded and its stage 2 in flight (the last stage's loads)
    v = item_n;
    nb = nb_n; img = img_n; X0 = X0_n; Y0 = Y0_n;
    nt32 = p.gb_off ? nb + grp * p.gb_off : nb * 2 + grp;
    ubase = ubase_nxt;
    tsel ^= 1;
    __syncthreads();                             // the exchange buffer sat in the V buffers
    produce(0, 0);
    __syncthreads();
#ifdef W4_TRACE
    if (P && threadIdx.x == 0 && v - v_step < 8192 && v - v_step >= 0) w4_trace2[(v - v_step) * 8 + 7] = __builtin_readcyclecounter();
#endif
  }
#undef W4_TAB
}

#ifdef W4_TRACE
extern "C" int csg_wino4_trace_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w4_trace), (size_t)n * 8);
}
extern "C" int csg_wino4_trace2_read(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(w4_trace2), (size_t)n * 8);
}
#endif

// ------------------------------------------------------------------------------------ host side
// T = 4: F(4x4,3x3), pad 1, output = input size.  T = 3: F(3x3,4x4), pad 1 or 2, output = input + 2 pad - 3.
static int w4_plan(const csg_wino_desc* d, int T, int pad, Wino4Params& p, size_t& shm, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  if (T == 4)
    CSG_REQUIRE(d->H % 4 == 0 && d->W % 4 == 0 && d->W >= 32 && d->H >= 16, CSG_E_UNSUPPORTED,
                "%s: F(4x4,3x3) needs H=%d, W=%d multiples of 4, W >= 32, H >= 16", who, d->H, d->W);
  else
    CSG_REQUIRE((pad == 1 || pad == 2) && d->H + 2 * pad >= 4 && d->W + 2 * pad >= 4, CSG_E_UNSUPPORTED,
                "%s: F(3x3,4x4) serves pad 1 or 2 (got %d) on maps of at least 4 - 2 pad pixels", who, pad);
  const int Ho = T == 4 ? d->H : d->H + 2 * pad - 3, Wo = T == 4 ? d->W : d->W + 2 * pad - 3;
  CSG_REQUIRE(d->Cin % 8 == 0 && d->x_cs % 4 == 0 && d->x_cs >= d->Cin && d->Cout % 4 == 0 && d->y_cs % 4 == 0 &&
                  d->y_cs >= d->Cout,
              CSG_E_UNSUPPORTED, "%s: Cin must be a multiple of 8, the other channel counts and strides of 4", who);
  CSG_REQUIRE((int64_t)d->B * (d->H + 1) * (d->W + 1) * (int64_t)(d->x_cs > d->y_cs ? d->x_cs : d->y_cs) * 4 < CSG_MAX_RECORDS,
              CSG_E_UNSUPPORTED, "%s: tensor too large for 32-bit byte offsets", who);
  CSG_REQUIRE((int64_t)36 * ((d->Cout + 31) / 32) * ((d->Cin + 7) / 8) * 1024 < CSG_MAX_RECORDS, CSG_E_UNSUPPORTED,
              "%s: packed weights too large for 32-bit byte offsets", who);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.Cout = d->Cout; p.y_cs = d->y_cs;
  p.Ho = Ho; p.Wo = Wo; p.pad = pad;
  p.tbx = (Wo + T * W4_TW - 1) / (T * W4_TW);
  p.tby = (Ho + T * W4_TH - 1) / (T * W4_TH);
  p.nblocks = (d->Cout + 63) / 64;
  p.NT32 = (d->Cout + 31) / 32;
  p.Q8 = (d->Cin + 7) / 8;
  p.act = d->act; p.slope = d->slope; p.gate_slope = 0.f;
  p.nt_off = 0; p.mod = 0; p.g_cs = 0; p.gb_off = 0; p.mod_slope = 1.f; p.mod_mean = nullptr; p.mod_invstd = nullptr;
  p.nstage = d->Cin / W4_PS;
  p.ksplit = 1;
  p.sps = p.nstage;
  p.slab = (long long)d->B * Ho * Wo * d->y_cs;
  p.nitems = 0;
  shm = (size_t)((T == 4 ? W4Geo<4>::OFFTAB : W4Geo<3>::OFFTAB) + W4_THREADS * 4) * 4;
  return CSG_OK;
}

// the launch shared by both tile sizes
template <int T, bool P>
static int w4_set_lds_limit(const char* who) {
  static DeviceOnce attr_once;
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (attr_once.pending(dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wino4_conv_v<T, P>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (P ? 160 : 128) * 1024);
    CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "%s: cannot raise the dynamic LDS limit: %s", who, hipGetErrorString(e));
    attr_once.mark(dev);
  }
  return CSG_OK;
}

// blocks of the persistent form: one per CU, a multiple of 8 (one share per XCD); 0 = the launch does not qualify.
// CSG_WINO4_PERSIST=0 switches it off (A/B runs).
static std::atomic<int> w4_persist_on{-1};         // -1: CSG_WINO4_PERSIST (default 1) decides at the first launch
static int w4_persistent_blocks(const Wino4Params& p, int64_t items) {
  static std::atomic<int> cus[32];                 // CU count per device (0 = not asked yet, -1 = unusable)
  int on = w4_persist_on.load(std::memory_order_relaxed);
  if (on < 0) {
    on = getenv("CSG_WINO4_PERSIST") ? atoi(getenv("CSG_WINO4_PERSIST")) : 1;
    int expect = -1;
    w4_persist_on.compare_exchange_strong(expect, on);          // (csg_wino4_persistent() may have decided meanwhile)
    on = w4_persist_on.load(std::memory_order_relaxed);
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (!on || dev < 0) return 0;
  int ncu = dev < 32 ? cus[dev].load(std::memory_order_relaxed) : 0;
  if (ncu == 0) {
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu < 8) ncu = -1;
    if (dev < 32) cus[dev].store(ncu, std::memory_order_relaxed);
  }
  const int g = ncu > 0 ? (ncu & ~7) : 0;
  if (g == 0 || p.ksplit != 1 || (p.nstage & 1) || p.nstage < 4 || (p.gb_off == 0 && (p.Cout & 63)) || items < 2 * (int64_t)g) return 0;
#ifdef W4_TRACE
  if (on == 2 && (items & 7) == 0) return (int)items;   // experiment: the persistent code, one item per block
#endif
  return g;
}

template <int T>
static int w4_launch(Wino4Params& p, size_t shm, int kid, double flops, const float* x, const float* packed,
                     const float* bias, const float* residual, const float* gate, float* y, float* workspace,
                     float* y_final, hipStream_t s, const char* who) {
  CSG_REQUIRE(shm <= 128 * 1024, CSG_E_UNSUPPORTED, "%s: %zu bytes of LDS", who, shm);
  const int64_t grid = (int64_t)p.B * p.tby * p.tbx * p.nblocks * p.ksplit;
  CSG_REQUIRE(grid < (1ll << 31), CSG_E_UNSUPPORTED, "%s: grid too large", who);
  p.nitems = (int)grid;
  const int pblocks = T == 4 ? w4_persistent_blocks(p, grid) : 0;
  int rc = pblocks ? w4_set_lds_limit<4, true>(who) : w4_set_lds_limit<T, false>(who);
  if (rc) return rc;
  ProfScope ps(kid, flops, s);
  if (pblocks)
    CSG_LAUNCH((k_wino4_conv_v<4, true>), dim3((unsigned)pblocks), dim3(W4_THREADS), shm + W4_THREADS * 16, s, p, x,
               (const float4*)packed, bias, residual, gate, y);
  else
    CSG_LAUNCH((k_wino4_conv_v<T, false>), dim3((unsigned)grid), dim3(W4_THREADS), shm, s, p, x, (const float4*)packed, bias,
               residual, gate, y);
  rc = check_launch(who);
  if (rc == CSG_OK && p.ksplit > 1) {
    launch_slab_reduce(workspace, p.slab, y_final, nullptr, 0, nullptr, p.ksplit, s);
    rc = check_launch("csg_wino4_conv(slab sum)");
  }
  return rc;
}

// Split over the input channels when the tile grid alone cannot fill the chip (see wn_split_plan in wino.hip)
static void w4_split_plan(Wino4Params& p, bool plain) {
  if (!plain || p.y_cs != p.Cout) return;
  const int64_t blocks = (int64_t)p.B * p.tby * p.tbx * p.nblocks;
  if (blocks >= 384 || p.nstage < 32) return;
  int ks = (int)((512 + blocks - 1) / blocks);
  if (ks > p.nstage / 16) ks = p.nstage / 16;
  if (ks < 2) return;
  p.sps = (p.nstage + ks - 1) / ks;
  p.ksplit = (p.nstage + p.sps - 1) / p.sps;
}

extern "C" {

int32_t csg_wino4_supported(const csg_wino_desc* d) {
  static const int on = getenv("CSG_WINO4") ? atoi(getenv("CSG_WINO4")) : 1;
  if (!on || d == nullptr) return 0;
  return (d->H % 4 == 0 && d->W % 4 == 0 && d->W >= 32 && d->H >= 16 && d->Cin % 8 == 0 && d->Cout % 4 == 0) ? 1 : 0;
}

int csg_wino4_pack_multi_launch(const PackMulti* pm, int blocks, double bytes, hipStream_t s) {
  ProfScope ps(K_WINO_PACK, bytes, s);
  CSG_LAUNCH(k_wino4_pack_multi, dim3((unsigned)blocks), dim3(256), 0, s, *pm);
  return check_launch("csg_wino_pack_weights_multi(F(4x4,3x3))");
}

int csg_wino_pack_weights_multi(const csg_wino_pack_item* items, int32_t n, void* stream) {
  CSG_REQUIRE(n >= 0 && (items != nullptr || n == 0), CSG_E_BADSHAPE, "csg_wino_pack_weights_multi: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  for (int variant = 2; variant <= 4; variant += 2) {
    PackMulti pm;
    int cnt = 0, blocks = 0;
    double bytes = 0.0;
    auto flush = [&]() -> int {
      if (cnt == 0) return CSG_OK;
      pm.n = cnt;
      const int rc = variant == 4 ? csg_wino4_pack_multi_launch(&pm, blocks, bytes, s)
                                  : csg_wino2_pack_multi_launch(&pm, blocks, bytes, s);
      cnt = 0;
      blocks = 0;
      bytes = 0.0;
      return rc;
    };
    for (int i = 0; i < n; ++i) {
      const csg_wino_pack_item& it = items[i];
      CSG_REQUIRE(it.variant == 2 || it.variant == 4, CSG_E_UNSUPPORTED,
                  "csg_wino_pack_weights_multi: item %d: variant %d (2 = F(2x2,3x3), 4 = F(4x4,3x3))", i, it.variant);
      if (it.variant != variant) continue;
      CSG_REQUIRE(it.w != nullptr && it.packed != nullptr && it.Cout > 0 && it.Cin > 0 && ((uintptr_t)it.packed % 16) == 0,
                  CSG_E_BADSHAPE, "csg_wino_pack_weights_multi: item %d: bad arguments", i);
      const int64_t N = it.backward_data ? it.Cin : it.Cout, K = it.backward_data ? it.Cout : it.Cin;
      const int64_t s_n = it.backward_data ? it.s_i : it.s_o, s_k = it.backward_data ? it.s_o : it.s_i;
      const int64_t span = (it.Cout - 1) * it.s_o + (it.Cin - 1) * it.s_i + 2 * it.s_h + 2 * it.s_w;
      CSG_REQUIRE(span < (1ll << 31) && s_n >= 0 && s_k >= 0 && it.s_h >= 0 && it.s_w >= 0, CSG_E_UNSUPPORTED,
                  "csg_wino_pack_weights_multi: item %d: weight too large for 32-bit element offsets", i);
      const int NT32 = (int)cdiv(N, 32), Q8 = (int)cdiv(K, 8);
      const int nb = (int)cdiv(Q8, 4) * NT32;
      if (cnt == CSG_PACK_MULTI || blocks + nb > (1 << 20)) {
        const int rc = flush();
        if (rc) return rc;
      }
      PackMultiItem& d = pm.it[cnt++];
      d.w = it.w; d.up = (float4*)it.packed;
      d.s_n = (int)s_n; d.s_k = (int)s_k; d.s_h = (int)it.s_h; d.s_w = (int)it.s_w;
      d.flip = it.backward_data ? 1 : 0; d.N = (int)N; d.K = (int)K; d.NT32 = NT32; d.Q8 = Q8;
      d.start = blocks;
      blocks += nb;
      bytes += (double)it.Cout * it.Cin * 9 * 4 + (double)(variant == 4 ? 36 : 16) * NT32 * Q8 * 64 * 16;
    }
    const int rc = flush();
    if (rc) return rc;
  }
  return CSG_OK;
}

int32_t csg_wino4_persistent(int32_t on) {
  const int prev = w4_persist_on;
  if (on >= 0) w4_persist_on = on != 0;
  return prev;
}

int64_t csg_wino4_pack_bytes(int64_t N, int64_t K) {
  if (N <= 0 || K <= 0) return -1;
  return (int64_t)36 * cdiv(N, 32) * cdiv(K, 8) * 64 * 16;
}

int csg_wino4_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                           int32_t backward_data, const float* sigma, float* packed, void* stream) {
  CSG_REQUIRE(w != nullptr && packed != nullptr && Cout > 0 && Cin > 0, CSG_E_BADSHAPE, "csg_wino4_pack_weights: bad arguments");
  CSG_REQUIRE(((uintptr_t)packed % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino4_pack_weights: packed must be 16-byte aligned");
  const int64_t N = backward_data ? Cin : Cout, K = backward_data ? Cout : Cin;
  const int64_t s_n = backward_data ? s_i : s_o, s_k = backward_data ? s_o : s_i;
  const int NT32 = (int)cdiv(N, 32), Q8 = (int)cdiv(K, 8);
  const int64_t total = (int64_t)36 * NT32 * Q8 * 64;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(K_WINO_PACK, (double)Cout * Cin * 9 * 4 + (double)total * 16, s);
  CSG_LAUNCH(k_wino4_pack<3>, dim3((unsigned)cdiv(Q8, 4), (unsigned)NT32), dim3(256), 0, s, w, s_n, s_k, s_h, s_w,
                     backward_data ? 1 : 0, (int)N, (int)K, sigma, NT32, Q8, (float4*)packed);
  return check_launch("csg_wino4_pack_weights");
}

int csg_wino34_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                            int32_t backward_data, const float* sigma, float* packed, void* stream) {
  CSG_REQUIRE(w != nullptr && packed != nullptr && Cout > 0 && Cin > 0, CSG_E_BADSHAPE, "csg_wino34_pack_weights: bad arguments");
  CSG_REQUIRE(((uintptr_t)packed % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino34_pack_weights: packed must be 16-byte aligned");
  const int64_t N = backward_data ? Cin : Cout, K = backward_data ? Cout : Cin;
  const int64_t s_n = backward_data ? s_i : s_o, s_k = backward_data ? s_o : s_i;
  const int NT32 = (int)cdiv(N, 32), Q8 = (int)cdiv(K, 8);
  const int64_t total = (int64_t)36 * NT32 * Q8 * 64;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(K_WINO_PACK, (double)Cout * Cin * 16 * 4 + (double)total * 16, s);
  CSG_LAUNCH(k_wino4_pack<4>, dim3((unsigned)cdiv(Q8, 4), (unsigned)NT32), dim3(256), 0, s, w, s_n, s_k, s_h, s_w,
                     backward_data ? 1 : 0, (int)N, (int)K, sigma, NT32, Q8, (float4*)packed);
  return check_launch("csg_wino34_pack_weights");
}

int32_t csg_wino34_supported(const csg_wino_desc* d, int32_t pad) {
  static const int on = getenv("CSG_WINO34") ? atoi(getenv("CSG_WINO34")) : 1;
  if (!on || d == nullptr) return 0;
  // the tile grid of 24 x 12 output pixels per block should not be mostly padding: maps from 16 pixels up
  return ((pad == 1 || pad == 2) && d->H >= 15 && d->W >= 15 && d->Cin % 8 == 0 && d->Cout % 4 == 0 && d->Cin >= 64 &&
          d->Cout >= 32) ? 1 : 0;
}

int64_t csg_wino34_conv_workspace(const csg_wino_desc* d, int32_t pad) {
  Wino4Params p;
  size_t shm = 0;
  if (w4_plan(d, 3, pad, p, shm, "csg_wino34_conv_workspace")) return -1;
  w4_split_plan(p, d->act == CSG_ACT_NONE);
  return p.ksplit > 1 ? (int64_t)p.ksplit * p.slab * 4 : 0;
}

int csg_wino34_conv(const csg_wino_desc* d, int32_t pad, const float* x, const float* packed, const float* bias,
                    const float* residual, const float* gate, float gate_slope, float* y, float* workspace,
                    int64_t workspace_bytes, void* stream) {
  Wino4Params p;
  size_t shm = 0;
  int rc = w4_plan(d, 3, pad, p, shm, "csg_wino34_conv");
  if (rc) return rc;
  w4_split_plan(p, d->act == CSG_ACT_NONE && bias == nullptr && residual == nullptr && gate == nullptr);
  p.gate_slope = gate_slope;
  if (p.ksplit > 1 && (workspace == nullptr || workspace_bytes < (int64_t)p.ksplit * p.slab * 4)) {   // no slabs: unsplit
    p.ksplit = 1;
    p.sps = p.nstage;
  }
  float* const y_final = y;
  if (p.ksplit > 1) {
    CSG_REQUIRE(((uintptr_t)workspace % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino34_conv: workspace must be 16-byte aligned");
    y = workspace;
  }
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                  (gate == nullptr || ((uintptr_t)gate % 16) == 0),
              CSG_E_UNSUPPORTED, "csg_wino34_conv: pointers must be 16-byte aligned");
  // algorithmic FLOPs of the DIRECT convolution this replaces (2 * M * 16*Cin * Cout)
  return w4_launch<3>(p, shm, K_WINO4_CONV, 2.0 * p.B * p.Ho * p.Wo * 16.0 * p.Cin * p.Cout, x, packed, bias, residual, gate,
                      y, workspace, y_final, (hipStream_t)stream, "csg_wino34_conv");
}

int64_t csg_wino4_conv_workspace(const csg_wino_desc* d) {
  Wino4Params p;
  size_t shm = 0;
  if (w4_plan(d, 4, 1, p, shm, "csg_wino4_conv_workspace")) return -1;
  w4_split_plan(p, d->act == CSG_ACT_NONE);
  return p.ksplit > 1 ? (int64_t)p.ksplit * p.slab * 4 : 0;
}

int csg_wino4_conv(const csg_wino_desc* d, const float* x, const float* packed, const float* bias, const float* residual,
                   const float* gate, float gate_slope, float* y, float* workspace, int64_t workspace_bytes, void* stream) {
  Wino4Params p;
  size_t shm = 0;
  int rc = w4_plan(d, 4, 1, p, shm, "csg_wino4_conv");
  if (rc) return rc;
  w4_split_plan(p, d->act == CSG_ACT_NONE && bias == nullptr && residual == nullptr && gate == nullptr);
  p.gate_slope = gate_slope;
  CSG_REQUIRE(gate == nullptr || ((uintptr_t)gate % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino4_conv: gate must be 16-byte aligned");
  if (p.ksplit > 1 && (workspace == nullptr || workspace_bytes < (int64_t)p.ksplit * p.slab * 4)) {   // no slabs: unsplit
    p.ksplit = 1;
    p.sps = p.nstage;
  }
  float* const y_final = y;
  if (p.ksplit > 1) {
    CSG_REQUIRE(((uintptr_t)workspace % 16) == 0, CSG_E_UNSUPPORTED, "csg_wino4_conv: workspace must be 16-byte aligned");
    y = workspace;
  }
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_wino4_conv: pointers must be 16-byte aligned");
  // algorithmic FLOPs of the DIRECT convolution this replaces (2 * M * 9*Cin * Cout): what FlopCounterMode counts
  return w4_launch<4>(p, shm, K_WINO4_CONV, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, x, packed, bias, residual, gate, y,
                      workspace, y_final, (hipStream_t)stream, "csg_wino4_conv");
}

int csg_wino4_conv_part(const csg_wino_desc* d, const float* x, const float* packed, int64_t tile_off, int64_t tiles_total,
                        const float* bias, const float* mod_x, const float* mod_gamma, int64_t gamma_cs,
                        const float* mod_mean, const float* mod_invstd, float mod_slope, float* y, void* stream) {
  Wino4Params p;
  size_t shm = 0;
  int rc = w4_plan(d, 4, 1, p, shm, "csg_wino4_conv_part");
  if (rc) return rc;
  CSG_REQUIRE(d->act == CSG_ACT_NONE, CSG_E_UNSUPPORTED, "csg_wino4_conv_part: no activation (the modulation has its own)");
  CSG_REQUIRE(tile_off >= 0 && tiles_total >= tile_off + p.NT32 && d->Cout % 32 == 0, CSG_E_BADSHAPE,
              "csg_wino4_conv_part: tiles [%ld, %ld) of %ld with Cout=%d", (long)tile_off, (long)(tile_off + p.NT32),
              (long)tiles_total, d->Cout);
  CSG_REQUIRE((int64_t)36 * tiles_total * p.Q8 * 1024 < CSG_MAX_RECORDS, CSG_E_UNSUPPORTED,
              "csg_wino4_conv_part: packed weights too large for 32-bit byte offsets");
  p.nt_off = (int)tile_off;
  p.NT32 = (int)tiles_total;                       // the stride of the packed operand
  if (mod_x != nullptr) {
    CSG_REQUIRE(mod_gamma != nullptr && mod_mean != nullptr && mod_invstd != nullptr && gamma_cs >= d->Cout &&
                    gamma_cs % 4 == 0,
                CSG_E_BADSHAPE, "csg_wino4_conv_part: the modulation needs gamma, mean, invstd");
    CSG_REQUIRE(((uintptr_t)mod_x % 16) == 0 && ((uintptr_t)mod_gamma % 16) == 0 && ((uintptr_t)mod_mean % 16) == 0 &&
                    ((uintptr_t)mod_invstd % 16) == 0,
                CSG_E_UNSUPPORTED, "csg_wino4_conv_part: pointers must be 16-byte aligned");
    p.mod = 1; p.g_cs = (int)gamma_cs; p.mod_slope = mod_slope; p.mod_mean = mod_mean; p.mod_invstd = mod_invstd;
  }
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                  (bias == nullptr || ((uintptr_t)bias % 16) == 0),
              CSG_E_UNSUPPORTED, "csg_wino4_conv_part: pointers must be 16-byte aligned");
  return w4_launch<4>(p, shm, K_WINO4_CONV, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, x, packed, bias, mod_x, mod_gamma, y,
                      nullptr, y, (hipStream_t)stream, "csg_wino4_conv_part");
}

// the joint launch exists in the persistent form only (w4_persistent_blocks: >= 2 items per CU, an even number of >= 4 stages)
static int w4_spade_plan(const csg_wino_desc* d, Wino4Params& p, size_t& shm, const char* who) {
  CSG_REQUIRE(d != nullptr && d->Cout % 32 == 0, CSG_E_UNSUPPORTED, "%s: C must be a multiple of 32", who);
  int rc = w4_plan(d, 4, 1, p, shm, who);
  if (rc) return rc;
  const int tiles = d->Cout / 32;
  p.NT32 = 2 * tiles;                              // the packed operand: gamma tiles then beta tiles
  p.nblocks = tiles;                               // a block = gamma tile nb + beta tile nb
  p.gb_off = tiles;
  p.mod = 2;
  const int64_t items = (int64_t)p.B * p.tby * p.tbx * p.nblocks;
  CSG_REQUIRE(items < (1ll << 31) && w4_persistent_blocks(p, items) > 0, CSG_E_UNSUPPORTED,
              "%s: the launch does not qualify for the persistent form (%ld items, %d stages)", who, (long)items, p.nstage);
  return CSG_OK;
}

int32_t csg_wino4_conv_spade_supported(const csg_wino_desc* d) {
  if (!csg_wino4_supported(d)) return 0;
  Wino4Params p;
  size_t shm = 0;
  const int rc = w4_spade_plan(d, p, shm, "csg_wino4_conv_spade_supported");
  return rc == CSG_OK ? 1 : 0;
}

int csg_wino4_conv_spade(const csg_wino_desc* d, const float* x, const float* packed, const float* bias, const float* mod_x,
                         float* gamma_out, int64_t gamma_cs, const float* mod_mean, const float* mod_invstd, float mod_slope,
                         float* y, void* stream) {
  Wino4Params p;
  size_t shm = 0;
  int rc = w4_spade_plan(d, p, shm, "csg_wino4_conv_spade");
  if (rc) return rc;
  CSG_REQUIRE(d->act == CSG_ACT_NONE, CSG_E_UNSUPPORTED, "csg_wino4_conv_spade: no activation (the modulation has its own)");
  CSG_REQUIRE(bias != nullptr && mod_x != nullptr && mod_mean != nullptr && mod_invstd != nullptr,
              CSG_E_BADSHAPE, "csg_wino4_conv_spade: needs the 2C biases, x, mean, invstd");
  CSG_REQUIRE(gamma_out == nullptr || (gamma_cs >= d->Cout && gamma_cs % 4 == 0 && ((uintptr_t)gamma_out % 16) == 0),
              CSG_E_BADSHAPE, "csg_wino4_conv_spade: bad gamma buffer");
  const int tiles = d->Cout / 32;
  CSG_REQUIRE((int64_t)36 * 2 * tiles * p.Q8 * 1024 < CSG_MAX_RECORDS, CSG_E_UNSUPPORTED,
              "csg_wino4_conv_spade: packed weights too large for 32-bit byte offsets");
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)packed % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                  ((uintptr_t)mod_x % 16) == 0 && ((uintptr_t)mod_mean % 16) == 0 && ((uintptr_t)mod_invstd % 16) == 0,
              CSG_E_UNSUPPORTED, "csg_wino4_conv_spade: pointers must be 16-byte aligned");
  p.g_cs = (int)gamma_cs; p.mod_slope = mod_slope; p.mod_mean = mod_mean; p.mod_invstd = mod_invstd;
  // algorithmic FLOPs of the direct gamma || beta convolution: 2C outputs
  return w4_launch<4>(p, shm, K_WINO4_CONV, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * 2.0 * p.Cout, x, packed, bias, mod_x, gamma_out,
                      y, nullptr, y, (hipStream_t)stream, "csg_wino4_conv_spade");
}

}  // extern "C"
