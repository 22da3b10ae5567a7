// K8w4g — weight gradient of a 3x3 / stride 1 / pad 1 convolution by Winograd F(3x3, 4x4) on the fp32 matrix cores of gfx950.
//
// Same call sites as k_wino_wgrad (wino.hip): the 3x3 nn.Conv2d layers of the generator — architecture.py:29-31 (conv_0 /
// conv_1), normalization.py:89-94 (mlp_gamma || mlp_beta) — on maps at least 32 pixels wide with >= 64 input channels:
// per 4x4 tile of dY and the 6x6 input patch around it,
//     dW += A^T [ (G dY_t G^T) (.) (B^T X_t B) ] A              36 multiplications per 16 output pixels
// where F(3x3,2x2) needs 64 and the direct sum 144.  Interpolation points {0, 1, -1, 1/2, -2, inf} (the set wino4.hip's
// header measures as the most accurate), B^T as in wino4.hip, G = rows c_k (1, p_k, p_k^2, p_k^3) with the scales c_k
// folded into the epilogue, A^T = [[1,1,1,1,1,0],[0,1,-1,1/2,-2,0],[0,1,1,1/4,4,1]].  The sum over tiles is taken INSIDE
// the brackets: 36 GEMMs  Acc_p[cout][cin] = sum_tiles E_p[tile][cout] V_p[tile][cin]  whose reduction index is the tile
// (the k = 2 of v_mfma_f32_32x32x2_f32 is a pair of horizontally adjacent tiles).
//
// Why it is shaped like this.  BOTH operands are transforms per (tile, channel), and on this chip VALU
// work does not hide under the fp32 MFMA (wino4.hip): what counts is VALU per MFMA.  A wave that forms its operands in
// registers pays ~10 VALU per MFMA (a 32x32 accumulator reuses each operand once); here every transform is computed ONCE
// per block and handed over through LDS in MFMA operand order, as wino4.hip does for V:
//   * a block owns 64 cout x 64 cin x HALF the positions (rows xi in {0,1,2} or {3,4,5} of the 6x6 transformed domain:
//     18 positions = 72 accumulators of 32x32, nine per wave, eight waves, two per SIMD).  Halving the positions instead
//     of the channels keeps the operand buffers inside LDS (36.9 KB per stage, double-buffered, beside the double-buffered
//     raw stage) AND makes a transform cheaper: only three rows of each operand are formed (the second block of the pair
//     forms the other three), ~4.1 VALU per MFMA.  The two halves are two slabs of the ordered slab sum: the output
//     transform is linear in the positions.
//   * stage = 2 x 2 tiles (8 x 8 output pixels): its 10 x 10 x 64-channel input patch and 8 x 8 x 64 dY pixels are staged by
//     LDS-DMA (buffer_load_dwordx4 ... lds; halo outside the image and channel tails are written as zeros).  Waves 0-3
//     each form V = rows xi of B^T X B for one (cin group, tile pair) — 6 x 6 raw reads, ~100 VALU, 18 operand stores;
//     waves 4-7 each form E' = rows xi of G' dY G'^T for one (cout group, tile pair) — 16 raw reads, ~50 VALU, 18 stores.
//     (Waves w and w + 4 share a SIMD: every SIMD carries one wave of either role.)
//   * one stage later every wave runs its 18 MFMAs out of the operand buffer (one ds_read_b64 per operand: both tile
//     pairs; a three-deep register ring fetched two positions ahead), one barrier per stage.  The two roles are template
//     parameters of the loop (their own register allocations: 208 VGPRs, no scratch); a launch is ONE wave of blocks (256).
//   * measured (tools/wgrad_bench.py, tools/wgrad_ablate.py): 1.2-1.4x over F(3x3,2x2) from 8 x 8 maps up; MFMA-only +
//     transform-only = everything (the fp32 MFMA and the VALU of a SIMD share their lanes), matrix pipe 52 % busy.
//   * the bias gradient is free: G' row 1 is (1,1,1,1), so E'[1][1] IS the sum of the dY tile.
//   * epilogue: accumulators through LDS (two rounds of 36), every thread applies A^T (.) A with the scales of G folded in
//     to its (cout, cin) pairs and writes the 3 x 3 taps of its slab; slabs are summed in order by k_slab_reduce
//     (bit-reproducible, no atomics).
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "csg_buffer.h"
#include "csg_common.h"
#include "csg_reduce.h"

using namespace csg;

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WW_THREADS 512
#define WW_XW 6400                       // floats of one raw X buffer: 10 x 10 pixels x 64 channels (25 DMA rounds of 64 lanes)
#define WW_YW 4096                       // floats of one raw dY buffer: 8 x 8 pixels x 64 channels (16 rounds)
#define WW_OPH 4608                      // floats of one operand array: 18 positions x 2 channel groups x 64 lanes x 2 tile pairs
#define WW_X0 0
#define WW_Y0 (2 * WW_XW)
#define WW_OP0 (2 * WW_XW + 2 * WW_YW)   // operand buffer b: E' at WW_OP0 + b * 2 * WW_OPH, V behind it
#define WW_LDS_FLOATS (WW_OP0 + 4 * WW_OPH)
#define WW_EXR 33                        // words per row of the epilogue exchange buffer

struct Wino4WgParams {
  int B, H, W, Cin, x_cs, Cout, y_cs;
  int RXn, RYn, nregions;        // 8 x 8-pixel stage regions per row / column of an image, in all
  int cblocks, kblocks;          // 64-channel blocks of cout / cin
  int nsplit, rps;               // slices of the regions, regions per slice
  int dbg;                       // developer ablations (CSG_WW_DBG bit mask): 1 no MFMAs, 2 no transform, 4 no DMA in the loop, 8 staggered E' waves
};

__device__ __forceinline__ int ww_xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// rows xi in {3 PH .. 3 PH + 2} of B^T t for one column of six values (B^T in wino4.hip)
template <int PH>
__device__ __forceinline__ void ww_bt_rows(const float (&t)[6], float (&o)[3]) {
  const float s31 = t[3] - t[1];
  if (PH == 0) {
    o[0] = fmaf(1.5f, s31, fmaf(-2.0f, t[2], t[0] + t[4]));
    o[1] = fmaf(2.5f, t[3], fmaf(0.5f, t[2], t[4] - t[1]));
    o[2] = fmaf(0.5f, t[3], fmaf(-2.5f, t[2], t[4] + t[1]));
  } else {
    const float s42 = t[4] - t[2];
    o[0] = fmaf(2.0f, s31, s42);
    o[1] = fmaf(-0.5f, s31, s42);
    o[2] = fmaf(1.5f, s42, fmaf(-2.0f, t[3], t[1] + t[5]));
  }
}
// all six rows
__device__ __forceinline__ void ww_bt_all(const float (&t)[6], float (&v)[6]) {
  const float s42 = t[4] - t[2], s31 = t[3] - t[1];
  v[0] = fmaf(1.5f, s31, fmaf(-2.0f, t[2], t[0] + t[4]));
  v[1] = fmaf(2.5f, t[3], fmaf(0.5f, t[2], t[4] - t[1]));
  v[2] = fmaf(0.5f, t[3], fmaf(-2.5f, t[2], t[4] + t[1]));
  v[3] = fmaf(2.0f, s31, s42);
  v[4] = fmaf(-0.5f, s31, s42);
  v[5] = fmaf(1.5f, s42, fmaf(-2.0f, t[3], t[1] + t[5]));
}
// G' = rows (1, p, p^2, p^3) of the points, unscaled, the point at infinity last: rows xi in {3 PH .. 3 PH + 2} / all six
template <int PH>
__device__ __forceinline__ void ww_g_rows(const float (&d)[4], float (&o)[3]) {
  if (PH == 0) {
    const float a = d[0] + d[2], b = d[1] + d[3];
    o[0] = d[0];
    o[1] = a + b;
    o[2] = a - b;
  } else {
    o[0] = fmaf(0.125f, d[3], fmaf(0.25f, d[2], fmaf(0.5f, d[1], d[0])));
    o[1] = fmaf(-8.0f, d[3], fmaf(4.0f, d[2], fmaf(-2.0f, d[1], d[0])));
    o[2] = d[3];
  }
}
__device__ __forceinline__ void ww_g_all(const float (&d)[4], float (&o)[6]) {
  const float a = d[0] + d[2], b = d[1] + d[3];
  o[0] = d[0];
  o[1] = a + b;
  o[2] = a - b;
  o[3] = fmaf(0.125f, d[3], fmaf(0.25f, d[2], fmaf(0.5f, d[1], d[0])));
  o[4] = fmaf(-8.0f, d[3], fmaf(4.0f, d[2], fmaf(-2.0f, d[1], d[0])));
  o[5] = d[3];
}

typedef __attribute__((address_space(3))) void* ww_lds_ptr;

// PH: the block's half of the positions; ISV: the wave's transform role (a template parameter: each role gets its own loop
// and its own register allocation — under a runtime branch the two roles' live ranges were merged and spilled)
template <int PH, bool ISV, bool STAG>
__device__ __forceinline__ void ww_body(const Wino4WgParams& p, const float* __restrict__ x, const float* __restrict__ dy,
                                        float* __restrict__ slabs, float* __restrict__ dbslabs, float* smem, int cb, int kb,
                                        int sp) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int r0 = sp * p.rps, r1 = min(p.nregions, r0 + p.rps);
  const int nst = r1 - r0;

  // ---- staging plan (stage-invariant).  Piece e = 64 round + lane of the linear [pixel][64 channels] image lands at LDS
  // byte 16 e.  X rounds of this wave: wave, wave + 8, wave + 16 (+ 24 for wave 0); dY rounds: wave, wave + 8.
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(x - (int64_t)(p.W + 1) * p.x_cs), 0, (int)(((long long)p.B * p.H * p.W + p.W + 1) * p.x_cs * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((long long)p.B * p.H * p.W * p.y_cs * 4), 0x00020000);
  unsigned xoff[4], yoff[2];
  int xflags = 0;                                // halo flags, 4 bits per round: 1 first row, 2 last row, 4 first column, 8 last column
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = (wave + 8 * i) * 64 + lane;
    const int pix = e >> 4, c4 = e & 15;
    const int row = pix / 10, col = pix - row * 10;
    const int ch = kb * 64 + c4 * 4;
    xoff[i] = (pix < 100 && ch < p.Cin) ? (unsigned)((row * p.W + col) * p.x_cs + ch) * 4u : CSG_OOB_OFF;
    xflags |= ((row == 0 ? 1 : 0) | (row == 9 ? 2 : 0) | (col == 0 ? 4 : 0) | (col == 9 ? 8 : 0)) << (4 * i);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = (wave + 8 * i) * 64 + lane;
    const int pix = e >> 4, c4 = e & 15;
    const int row = pix >> 3, col = pix & 7;
    const int ch = cb * 64 + c4 * 4;
    yoff[i] = ch < p.Cout ? (unsigned)((row * p.W + col) * p.y_cs + ch) * 4u : CSG_OOB_OFF;
  }
  // region of the next stage to fetch, walked incrementally (no divisions in the loop): (rx, ry, img) of region r0 + dma_s
  int dma_s = 0, dma_rx, dma_ry, dma_img;
  {
    const int rr = min(r0, p.nregions - 1);
    dma_rx = rr % p.RXn;
    const int t = rr / p.RXn;
    dma_ry = t % p.RYn;
    dma_img = t / p.RYn;
  }
  auto dma_stage = [&](int bufsel) {             // the next stage of this block -> raw buffer bufsel; past the end: the last
    const int y0 = dma_ry * 8, x0 = dma_rx * 8;  // region again (never consumed)
    const int edge = (y0 == 0 ? 1 : 0) | (y0 + 8 == p.H ? 2 : 0) | (x0 == 0 ? 4 : 0) | (x0 + 8 == p.W ? 8 : 0);
    const int pix0 = (dma_img * p.H + y0) * p.W + x0;
    float* bx = smem + WW_X0 + bufsel * WW_XW + wave * 256;
    float* by = smem + WW_Y0 + bufsel * WW_YW + wave * 256;
    if (edge == 0) {                             // interior region (most of a large map): no halo lane to mask, no VALU at all
#pragma unroll
      for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ww_lds_ptr)(bx + i * 2048), 16, (int)xoff[i], pix0 * p.x_cs * 4, 0, 0);
      if (wave == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ww_lds_ptr)(bx + 3 * 2048), 16, (int)xoff[3], pix0 * p.x_cs * 4, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ww_lds_ptr)(bx + i * 2048), 16,
                                                 (int)(((xflags >> (4 * i)) & edge) ? CSG_OOB_OFF : xoff[i]), pix0 * p.x_cs * 4, 0, 0);
      if (wave == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (ww_lds_ptr)(bx + 3 * 2048), 16,
                                                 (int)(((xflags >> 12) & edge) ? CSG_OOB_OFF : xoff[3]), pix0 * p.x_cs * 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (ww_lds_ptr)(by + i * 2048), 16, (int)yoff[i], pix0 * p.y_cs * 4, 0, 0);
    if (r0 + dma_s + 1 < p.nregions) {           // advance (the last region of the tensor repeats)
      ++dma_s;
      if (++dma_rx == p.RXn) {
        dma_rx = 0;
        if (++dma_ry == p.RYn) {
          dma_ry = 0;
          ++dma_img;
        }
      }
    }
  };

  // ---- roles.  Transform: waves 0-3 form V for (cin group, tile pair) = (wave & 1, wave >> 1), waves 4-7 form E' for
  // (cout group, tile pair) likewise; lane = (channel c of the group, tile h of the pair).  MFMA: wave w owns the channel
  // groups (mt, nt) = ((w >> 1) & 1, w & 1) at the local positions pl = 2 k + (w >> 2), k = 0..8.
  constexpr bool is_v = ISV;
  const int grp = wave & 1, pr = (wave >> 1) & 1;
  const int mt = (wave >> 1) & 1, nt = wave & 1, plo = wave >> 2;
  // raw words of this lane's tile: X patch rows 4 pr + i, columns 4 h + j; dY rows 4 pr + i, columns 4 h + j
  const float* rawx = smem + WW_X0 + ((4 * pr) * 10 + 4 * h) * 64 + grp * 32 + c;
  const float* rawy = smem + WW_Y0 + ((4 * pr) * 8 + 4 * h) * 64 + grp * 32 + c;
  // operand words (transform role: this lane's slot of local position 0; the tile pair pr is the low index)
  float* opw = smem + WW_OP0 + (is_v ? WW_OPH : 0) + (grp * 64 + lane) * 2 + pr;
  // operand words (MFMA role), float2 = both tile pairs
  const float* ope = smem + WW_OP0 + ((plo * 2 + mt) * 64 + lane) * 2;
  const float* opv = smem + WW_OP0 + WW_OPH + ((plo * 2 + nt) * 64 + lane) * 2;

  const bool do_db = dbslabs != nullptr && kb == 0 && PH == 0 && !is_v;
  float dbacc = 0.f;

  f32x16 acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

  // ---- one stage in four phases, kept apart by scheduling barriers so that no LDS latency sits in front of an MFMA or a
  // transform instruction: (1) the raw reads of the transform of stage s + 1 and the operand reads of the MFMAs of stage s are
  // issued together, (2) the 18 MFMAs run as soon as the operands are in, (3) the transform (its inputs arrived long ago) and
  // its operand stores, (4) the barrier.
  float raw[36];                                 // V role: the 6 x 6 patch [i][j]; E' role: the 4 x 4 dY tile [i][j] in raw[0..15]
  auto load_raw = [&](int rb) {
    if (is_v) {
      const float* src = rawx + rb * WW_XW;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) raw[i * 6 + j] = src[(i * 10 + j) * 64];
    } else {
      const float* src = rawy + rb * WW_YW;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) raw[i * 4 + j] = src[(i * 8 + j) * 64];
    }
  };
  // three rows xi of this block's half, into operand buffer ob; live: the stage exists (the bias sum must not count a
  // clamped one)
  auto transform = [&](int ob, bool live) {
    float* dst = opw + ob * (2 * WW_OPH);
    if (is_v) {
      float t[3][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float d[6] = {raw[j], raw[6 + j], raw[12 + j], raw[18 + j], raw[24 + j], raw[30 + j]};
        float o[3];
        ww_bt_rows<PH>(d, o);
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r][j] = o[r];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v[6];
        ww_bt_all(t[r], v);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) dst[(r * 6 + nu) * 256] = v[nu];
      }
    } else {
      float t[3][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d[4] = {raw[j], raw[4 + j], raw[8 + j], raw[12 + j]};
        float o[3];
        ww_g_rows<PH>(d, o);
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r][j] = o[r];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float v[6];
        ww_g_all(t[r], v);
        if (PH == 0 && r == 1 && do_db && live) dbacc += v[1];      // E'[1][1] = the sum of the 4 x 4 dY tile
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) dst[(r * 6 + nu) * 256] = v[nu];
      }
    }
  };
  // the 18 MFMAs of a stage out of operand buffer ob, the operands of position k + 2 fetched while position k multiplies
  // (a ring of three register pairs per operand: the whole set up front would not fit 256 registers beside the raw patch).
  // The first two positions are fetched BEFORE the raw reads of the transform (LDS answers in order: the first MFMA then waits
  // for its own operands only, not for the 16 - 36 raw words behind them).
  csg_f32x2 er[3], vr[3];
  auto ops_prefill = [&](int ob) {
    const float* pe = ope + ob * (2 * WW_OPH);
    const float* pv = opv + ob * (2 * WW_OPH);
    er[0] = *(const csg_f32x2*)(pe);
    vr[0] = *(const csg_f32x2*)(pv);
    er[1] = *(const csg_f32x2*)(pe + 512);
    vr[1] = *(const csg_f32x2*)(pv + 512);
  };
  auto mfmas = [&](int ob) {
    const float* pe = ope + ob * (2 * WW_OPH);
    const float* pv = opv + ob * (2 * WW_OPH);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      if (k + 2 < 9) {
        er[(k + 2) % 3] = *(const csg_f32x2*)(pe + (k + 2) * 512);
        vr[(k + 2) % 3] = *(const csg_f32x2*)(pv + (k + 2) * 512);
      }
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(er[k % 3].x, vr[k % 3].x, acc[k], 0, 0, 0);
      acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(er[k % 3].y, vr[k % 3].y, acc[k], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // the two reads of position k + 2 ...
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // ... then the two MFMAs of position k
    }
  };

  if (nst > 0) {
    dma_stage(0);
    dma_stage(1);
    __syncthreads();                             // (its fence waits for this wave's DMAs: vmcnt(0))
    load_raw(0);
    transform(0, true);
    __syncthreads();
    // stage s: DMA of stage s + 2 into the raw buffer stage s left; transform of stage s + 1; MFMAs of stage s
    auto stage = [&](int s, auto par_tag) {
      constexpr int par = decltype(par_tag)::value;
      if (!(p.dbg & 4)) dma_stage(par);
      if (ISV || !STAG) {
        if (!(p.dbg & 1)) ops_prefill(par);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2)) load_raw(par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 1)) mfmas(par);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2)) transform(par ^ 1, s + 1 < nst);
      } else {
        // STAG: the E' waves run the two halves of a stage in the OTHER order (a stagger: MI355X_MICROARCH.md, two waves per
        // SIMD, item 9), so that one partner of a SIMD has matrix work while the other transforms.  Measured equal to the
        // plain order (0.912 vs 0.901 ms, same box): MFMA-only + transform-only = everything either way — on this chip the
        // fp32 MFMA and the VALU of a SIMD share their lanes.  Off by default.
        if (!(p.dbg & 2)) load_raw(par ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2)) transform(par ^ 1, s + 1 < nst);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 1)) ops_prefill(par);
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 1)) mfmas(par);
      }
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    };
    int s = 0;
    for (; s + 1 < nst; s += 2) {
      stage(s, std::integral_constant<int, 0>());
      stage(s + 1, std::integral_constant<int, 1>());
    }
    if (s < nst) stage(s, std::integral_constant<int, 0>());
  }
  __syncthreads();

  // ---- epilogue.  dW[a][b] = sum over this half's positions of (A^T[a][xi] c_xi) (A^T[b][nu] c_nu) Acc'[xi][nu], c = the row
  // scales of G: (1, 1/3, -1/3, -16/15, 1/15, 1).  Two rounds of four waves (one mt each): the 36 accumulators of a round
  // in LDS as [nt][pl][32 cout][33], then every thread forms the nine taps of four (cout, cin) pairs.
  const float Atc[3][6] = {{1.0f, (float)(1.0 / 3.0), (float)(-1.0 / 3.0), (float)(-16.0 / 15.0), (float)(1.0 / 15.0), 0.0f},
                           {0.0f, (float)(1.0 / 3.0), (float)(1.0 / 3.0), (float)(-8.0 / 15.0), (float)(-2.0 / 15.0), 0.0f},
                           {0.0f, (float)(1.0 / 3.0), (float)(-1.0 / 3.0), (float)(-4.0 / 15.0), (float)(4.0 / 15.0), 1.0f}};
  float* slab = slabs + (int64_t)(sp * 2 + PH) * p.Cout * 9 * p.Cin;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (mt == round) {
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int pl = 2 * k + plo;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int i = (e & 3) + 8 * (e >> 2) + 4 * h;          // cout row inside the 32 x 32 quadrant
          smem[((nt * 18 + pl) * 32 + i) * WW_EXR + c] = acc[k][e];
        }
      }
    }
    __syncthreads();
    const int j = tid & 31;
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      const int q = (tid >> 5) + 16 * rep;                       // 0..63: (nt, cout row)
      const int ntq = q >> 5, i = q & 31;
      const int cout_g = cb * 64 + round * 32 + i, cin_g = kb * 64 + ntq * 32 + j;
      if (cout_g < p.Cout && cin_g < p.Cin) {
        float R[3][3];                                           // [local row xi][b] = sum_nu Atc[b][nu] M[xi][nu]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          float m[6];
#pragma unroll
          for (int nu = 0; nu < 6; ++nu) m[nu] = smem[((ntq * 18 + r * 6 + nu) * 32 + i) * WW_EXR + j];
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            float a = 0.f;
#pragma unroll
            for (int nu = 0; nu < 6; ++nu)
              if (Atc[b][nu] != 0.0f) a = fmaf(Atc[b][nu], m[nu], a);
            R[r][b] = a;
          }
        }
        float* dst = slab + (int64_t)cout_g * 9 * p.Cin + cin_g;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) {
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r)
              if (Atc[a][3 * PH + r] != 0.0f) v = fmaf(Atc[a][3 * PH + r], R[r][b], v);
            dst[(a * 3 + b) * p.Cin] = v;
          }
      }
    }
    __syncthreads();
  }
  // bias gradient: the E' waves of (cout group, tile pair) hold the sums of their tiles; the two tile pairs meet in LDS
  if (dbslabs != nullptr && kb == 0) {
    if (PH == 0) {
      if (!is_v) {
        const float tot = dbacc + __shfl_xor(dbacc, 32, 64);
        if (h == 0) smem[pr * 64 + grp * 32 + c] = tot;
      }
      __syncthreads();
      if (tid < 64 && cb * 64 + tid < p.Cout) dbslabs[(int64_t)(sp * 2) * p.Cout + cb * 64 + tid] = smem[tid] + smem[64 + tid];
    } else if (tid < 64 && cb * 64 + tid < p.Cout) {
      dbslabs[(int64_t)(sp * 2 + 1) * p.Cout + cb * 64 + tid] = 0.f;     // the other half's slab row takes part in the sum
    }
  }
}

template <bool STAG>
__global__ __launch_bounds__(WW_THREADS, 2) void k_wino4_wgrad(Wino4WgParams p, const float* __restrict__ x,
                                                               const float* __restrict__ dy, float* __restrict__ slabs,
                                                               float* __restrict__ dbslabs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int bid = ww_xcd_remap(blockIdx.x, gridDim.x);
  // the two position halves and all (cout block, cin block) pairs of one slice are adjacent: the slice's dY and X stay in
  // that XCD's L2
  const int ph = bid & 1;
  bid >>= 1;
  const int kb = bid % p.kblocks;
  bid /= p.kblocks;
  const int cb = bid % p.cblocks;
  const int sp = bid / p.cblocks;
  const bool is_v = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;
  if (ph == 0) {
    if (is_v)
      ww_body<0, true, STAG>(p, x, dy, slabs, dbslabs, smem, cb, kb, sp);
    else
      ww_body<0, false, STAG>(p, x, dy, slabs, dbslabs, smem, cb, kb, sp);
  } else {
    if (is_v)
      ww_body<1, true, STAG>(p, x, dy, slabs, dbslabs, smem, cb, kb, sp);
    else
      ww_body<1, false, STAG>(p, x, dy, slabs, dbslabs, smem, cb, kb, sp);
  }
}

static int ww_plan(const csg_wino_desc* d, Wino4WgParams& p, const char* who) {
  CSG_REQUIRE(d != nullptr, CSG_E_BADSHAPE, "%s: null descriptor", who);
  CSG_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, CSG_E_BADSHAPE, "%s: non-positive dimension", who);
  CSG_REQUIRE(d->W % 8 == 0 && d->H % 8 == 0, CSG_E_UNSUPPORTED, "%s: H=%d, W=%d must be multiples of the 8x8-pixel stage", who,
              d->H, d->W);
  CSG_REQUIRE(d->Cin % 4 == 0 && d->Cout % 4 == 0 && d->x_cs % 4 == 0 && d->y_cs % 4 == 0 && d->x_cs >= d->Cin &&
                  d->y_cs >= d->Cout,
              CSG_E_UNSUPPORTED, "%s: channel counts and strides must be multiples of 4", who);
  CSG_REQUIRE(((int64_t)d->B * d->H * d->W + d->W + 1) * (int64_t)(d->x_cs > d->y_cs ? d->x_cs : d->y_cs) * 4 < CSG_MAX_RECORDS,
              CSG_E_UNSUPPORTED, "%s: tensor too large for 32-bit byte offsets", who);
  p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cs = d->x_cs; p.Cout = d->Cout; p.y_cs = d->y_cs;
  p.RXn = d->W / 8;
  p.RYn = d->H / 8;
  const int64_t nr = (int64_t)d->B * p.RXn * p.RYn;
  CSG_REQUIRE(nr < (1ll << 30), CSG_E_UNSUPPORTED, "%s: too many regions", who);
  p.nregions = (int)nr;
  p.cblocks = (d->Cout + 63) / 64;
  p.kblocks = (d->Cin + 63) / 64;
  // one block per CU is resident: aim at ONE wave of blocks (256: every CU gets one block of equal length — measured 256 / 512 /
  // 768 / 1024: 4.92 / 5.00 / 5.13 / 5.24 ms over eight generator shapes, fewer slabs to sum and fewer epilogues), at least 16
  // stages per block, at most 256 slices
  static const int target = getenv("CSG_WINO4_WGRAD_BLOCKS") ? atoi(getenv("CSG_WINO4_WGRAD_BLOCKS")) : 256;
  const int tiles2d = p.cblocks * p.kblocks * 2;
  int ns = (target + tiles2d - 1) / tiles2d;
  const int max_ns = (int)((nr + 15) / 16);
  if (ns > max_ns) ns = max_ns;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  p.rps = (int)((nr + ns - 1) / ns);
  p.nsplit = (int)((nr + p.rps - 1) / p.rps);
  p.dbg = getenv("CSG_WW_DBG") ? atoi(getenv("CSG_WW_DBG")) : 0;      // (read per call: tools/wgrad_ablate.py flips it)
  return CSG_OK;
}

extern "C" {

int64_t csg_wino4_bwd_weight_workspace(const csg_wino_desc* d) {
  Wino4WgParams p;
  if (ww_plan(d, p, "csg_wino4_bwd_weight_workspace")) return -1;
  return (int64_t)p.nsplit * 2 * d->Cout * (9 * (int64_t)d->Cin + 1) * 4;
}

int csg_wino4_bwd_weight(const csg_wino_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                         int64_t workspace_bytes, void* stream) {
  Wino4WgParams p;
  int rc = ww_plan(d, p, "csg_wino4_bwd_weight");
  if (rc) return rc;
  const int64_t wsize = (int64_t)d->Cout * 9 * d->Cin;
  const int nslab = p.nsplit * 2;
  const int64_t need = (int64_t)nslab * (wsize + d->Cout) * 4;
  CSG_REQUIRE(workspace != nullptr && workspace_bytes >= need, CSG_E_WORKSPACE, "csg_wino4_bwd_weight: workspace %ld < %ld bytes",
              (long)workspace_bytes, (long)need);
  static DeviceOnce attr_once;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t ep_bytes = (size_t)2 * 18 * 32 * WW_EXR * 4;
  const size_t shm = (size_t)WW_LDS_FLOATS * 4 > ep_bytes ? (size_t)WW_LDS_FLOATS * 4 : ep_bytes;
  if (attr_once.pending(dev)) {
    for (const void* fn : {(const void*)k_wino4_wgrad<true>, (const void*)k_wino4_wgrad<false>}) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_wino4_bwd_weight: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e));
    }
    attr_once.mark(dev);
  }
  hipStream_t s = (hipStream_t)stream;
  float* dbslabs = db != nullptr ? workspace + (int64_t)nslab * wsize : nullptr;
  {
    ProfScope ps(K_WINO4_WGRAD, 2.0 * p.B * p.H * p.W * 9.0 * p.Cin * p.Cout, s);
    const dim3 grid((unsigned)(p.cblocks * p.kblocks * 2 * p.nsplit));
    if (p.dbg & 8)                               // the staggered form (measured equal: DESIGN.md 8) stays selectable
      CSG_LAUNCH(k_wino4_wgrad<true>, grid, dim3(WW_THREADS), shm, s, p, x, dy, workspace, dbslabs);
    else
      CSG_LAUNCH(k_wino4_wgrad<false>, grid, dim3(WW_THREADS), shm, s, p, x, dy, workspace, dbslabs);
    rc = check_launch("csg_wino4_bwd_weight");
    if (rc) return rc;
  }
  {
    ProfScope ps(K_WGRAD_REDUCE, (double)(nslab + 1) * wsize * 4, s);
    launch_slab_reduce(workspace, wsize, dw, dbslabs, db != nullptr ? d->Cout : 0, db, nslab, s);
    rc = check_launch("csg_wino4_bwd_weight(reduce)");
  }
  return rc;
}

}  // extern "C"
