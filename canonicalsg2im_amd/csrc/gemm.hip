// Plain fp32 GEMMs on the matrix cores for the layers that ARE matrix products: nn.Linear of the graph encoder
// (reference sg2im/graph.py:63-77 net1 / net2 through sg2im/layers.py build_mlp — at config C5 100 000 triplet rows
// through 384 -> 512 -> 1152) and 1x1 convolutions on NHWC maps (SPADEResnetBlock.conv_s, architecture.py:37-39).
// The implicit-GEMM convolution kernel (igemm.hip) serves them too, but pays for its generality there: a tap table, a
// magic division per staged row, register staging (global -> VGPR -> ds_write).  Here:
//
//   k_gemm_nt   Y[M][N] = epi(A[M][K] . Bw[N][K]^T)       forward (Bw = W) and backward-data (A = dY, Bw = W^T)
//   k_gemm_tn   dW[N][K] = sum_m dY[m][N]^T . X[m][K]      weight gradient (+ column sums of dY = bias gradient)
//
// Both: 128 x 128 output tiles, four waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 accumulators = 64 registers), two
// blocks per CU, two 32 KB LDS stages filled by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPRs, no ds_write, no VALU
// beside the fp32 MFMAs, which share their issue slot with it on this chip — DESIGN 4.1b).
// * nt: both operands have the reduction index contiguous.  A stage is 32 k = eight 16-byte slots per row; the DMA
//   writes 1 KB per wave-instruction linearly (8 rows), so the XOR swizzle that makes the ds_read_b128 of the MFMA
//   operands conflict-free is applied to the SOURCE address of each lane: LDS slot s of row r holds the row's k-slot
//   s ^ ((r >> 1) & 7) (rows are 128 B, a ds_read_b128 is served in groups of 16 lanes whose rows differ in exactly
//   those bits: cdna_hip_programming.md T2).  The reduction order inside a stage is free, so lane half h = lane / 32
//   takes k-slot 2g + h as its four MFMA steps of group g: one ds_read_b128 feeds four MFMAs per operand.
// * tn: the reduction index (rows m) is the strided one; tiles are staged row by row ([32 m][128 n], [32 m][128 k]) and
//   read with ds_read_b32 (32 consecutive floats per half-wave: conflict-free), one read per MFMA.  The rows are cut
//   into slices, one slab per slice, summed in a fixed order (bit-reproducible, csg_reduce.h).
#include <stdlib.h>

#include "csg_buffer.h"
#include "csg_common.h"
#include "csg_reduce.h"

using namespace csg;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr;

struct GemmParams {
  int M, N, K;
  int lda, ldb, ldy, ldg;   // floats per row of A, Bw, Y, gate
  int act;
  float slope;
  float gate_slope;
  int nnb;                  // column blocks of 128
  int nstage;               // ceil(K / 32)
};

struct GemmTnParams {
  int M, N, K;
  int ldy, ldx;             // floats per row of dY, X
  int nnb, nkb;             // 128-blocks along N (rows of dW) and K (columns)
  int nsplit, rows_per;     // row slices; rows per slice (a multiple of 32)
  int direct;               // 1: one slice, results go straight to dw / db
};

__device__ __forceinline__ int gm_xcd_remap(int bid, int nblk) {   // consecutive logical ids on ONE XCD (its own L2)
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

#define GM_STAGE_FLOATS 8192     // 128 x 32 (A) + 128 x 32 (B), or 32 x 128 + 32 x 128
#define GM_LDS_BYTES (2 * GM_STAGE_FLOATS * 4)

// ------------------------------------------------------------------------------------------------ Y = epi(A . Bw^T)
// NTW = 32-column accumulators per wave: 2 -> 128 x 128 tiles, 1 -> 128 x 64 (layers with at most 64 outputs: conv_s of
// the last residual block; a half-empty 128-wide tile would double their matrix work).
// BK = reduction depth of a stage (32 or 16), NBUF = LDS stages in the ring: the DMA of stage s + NBUF - 1 is issued at the top
// of stage s.  (BK, NBUF) = (32, 2): 64 KB, two blocks per CU; (16, 2): 32 KB, four blocks; (16, 3): 48 KB, three; (16, 4): 64 KB, two.
// Every stage issues the same number of DMAs (past the end of K all their lanes are out of range and write zeros into a
// buffer nobody reads), so the wait for "stage s + 1 has landed" is a constant vmcnt.
template <int NTW, int BK, int NBUF, bool GATED>
__global__ __launch_bounds__(256, GATED ? 2 : (BK == 16 ? 8 : 4) / NBUF) void k_gemm_nt(GemmParams p, const float* __restrict__ A,
                                                                   const float* __restrict__ Bw, const float* __restrict__ bias,
                                                                   const float* __restrict__ gate, float* __restrict__ Y) {
  constexpr int BN = 64 * NTW;
  constexpr int SLOTS = BK / 4;                       // 16-byte k-slots per row and stage
  constexpr int SH = BK == 32 ? 1 : 2;                // rows per 256-byte bank row = 2^SH
  constexpr int NA = 128 * SLOTS / 256, NB = BN * SLOTS / 256;   // DMA instructions per thread and stage
  constexpr int RPI = 256 / SLOTS;                    // rows one 256-lane DMA covers
  constexpr int ASZ = 128 * BK, STG = (128 + BN) * BK;           // floats: A part of a stage, whole stage
  constexpr int G = BK / 8;                           // k-groups per stage (one ds_read_b128 per operand and group)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int bid = gm_xcd_remap(blockIdx.x, gridDim.x);
  const int nb = bid % p.nnb, mb = bid / p.nnb;       // all column blocks of one row tile adjacent: A's tile stays in L2
  const int m0 = mb * 128, n0 = nb * BN;
  const int mrem = p.M - m0, nrem = p.N - n0;

  // descriptors rebased at the tile's first row: offsets stay far below 2^31 whatever M is; rows past the end of the
  // matrix lie beyond num_records (lda >= K) and read as zeros
  const float* Ab = A + (int64_t)m0 * p.lda;
  const float* Bb = Bw + (int64_t)n0 * p.ldb;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Ab, 0, (int)(((int64_t)(min(mrem, 128) - 1) * p.lda + p.K) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
      (void*)Bb, 0, (int)(((int64_t)(min(nrem, BN) - 1) * p.ldb + p.K) * 4), 0x00020000);

  // ---- staging plan: DMA i of this thread fills LDS slot q = tid + 256 i of the [rows][SLOTS] image, from row
  // q / SLOTS = tid / SLOTS + RPI i, k-slot (q % SLOTS) ^ ((row >> SH) & (SLOTS - 1)) — the same for every i
  const int row0 = tid / SLOTS;
  const int kslot = (tid & (SLOTS - 1)) ^ ((row0 >> SH) & (SLOTS - 1));
  unsigned offA[NA], offB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) offA[i] = (unsigned)((row0 + RPI * i) * p.lda + kslot * 4) * 4u;
#pragma unroll
  for (int i = 0; i < NB; ++i) offB[i] = (unsigned)((row0 + RPI * i) * p.ldb + kslot * 4) * 4u;
  auto dma_stage = [&](int s, int bufsel) {
    const int k0 = s * BK;
    const bool kin = kslot * 4 + k0 < p.K;            // K % 4 == 0: a slot is inside or outside as a whole
    float* base = smem + bufsel * STG + wave * 256;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(base + i * 1024), 16, (int)(kin ? offA[i] : CSG_OOB_OFF),
                                               k0 * 4, 0, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(base + ASZ + i * 1024), 16,
                                               (int)(kin ? offB[i] : CSG_OOB_OFF), k0 * 4, 0, 0);
  };

  // ---- operand reads: row R = 64 w + 32 t + c, k-slot 2 g + h, at LDS slot (2 g + h) ^ ((R >> SH) & (SLOTS - 1))
  const int sw = (c >> SH) & (SLOTS - 1);
  int ra[G], rb[G];                                   // float offsets of the k-groups, tile t = 0
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int slot = ((2 * g + h) ^ sw) * 4;
    ra[g] = (wm * 64 + c) * BK + slot;
    rb[g] = ASZ + (wn * 32 * NTW + c) * BK + slot;
  }
  struct Frag {
    csg_f32x4 a[2], b[NTW];
  };
  auto read_frag = [&](int bufsel, int g, Frag& f) {
    const float* s = smem + bufsel * STG;
    f.a[0] = *(const csg_f32x4*)(s + ra[g]);
    f.a[1] = *(const csg_f32x4*)(s + ra[g] + 32 * BK);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) f.b[nt] = *(const csg_f32x4*)(s + rb[g] + nt * 32 * BK);
  };

  f32x16 acc[2][NTW];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;

  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.b[nt][j], f.a[mt][j], acc[mt][nt], 0, 0, 0);
    }
  };

  constexpr int NDMA = NA + NB;
  // s_waitcnt vmcnt(n) only (expcnt / lgkmcnt fields at their maxima): simm16 = vmcnt[3:0] | 0x70 | 0xF00 | vmcnt[5:4] << 14
  constexpr int PENDING = NDMA * (NBUF - 2);           // DMAs that may still be in flight when stage s + 1 must have landed
  static_assert(PENDING < 64, "vmcnt is a 6-bit counter");
  constexpr int WAITIMM = (PENDING & 15) | 0x70 | 0xF00 | ((PENDING >> 4) << 14);
  auto stage_barrier = [&]() {                         // this wave's DMAs of the next stage landed; then everybody's
    __builtin_amdgcn_s_waitcnt(WAITIMM);
    __builtin_amdgcn_s_barrier();
  };
  Frag f[2];
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) dma_stage(i, i);
  stage_barrier();
  __builtin_amdgcn_sched_barrier(0);
  read_frag(0, 0, f[0]);
  int buf = 0, fill = NBUF - 1;                        // buffer of stage s; buffer the DMA of stage s + NBUF - 1 goes to
  for (int s = 0; s < p.nstage; ++s) {
    dma_stage(s + NBUF - 1, fill);                     // the buffer every wave finished reading before the last barrier
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g + 1 < G; ++g) {
      read_frag(buf, g + 1, f[(g + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);              // reads first: their latency under the MFMAs that follow
      compute(f[g & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int nxt = buf + 1 == NBUF ? 0 : buf + 1;
    stage_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < p.nstage) read_frag(nxt, 0, f[0]);     // first group of the next stage under the last MFMAs of this one
    compute(f[(G - 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    fill = buf;
    buf = nxt;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): no DMA of this block may land after its LDS is released

  // ---- epilogue.  The weight rows are the FIRST MFMA operand: D[i = n][j = m], so a lane holds one row m = c and, per
  // accumulator, four runs of four consecutive columns n = 8 q + 4 h + {0..3}: 16-byte stores (dword stores of the other
  // operand order: 64 store instructions per wave instead of 16).  EVERY load of the epilogue (bias, gate) is issued before
  // the first store: vmcnt retires in order, so a load behind a store makes its consumer wait for the store's write
  // acknowledge — with the loads interleaved the sixteen stores of a wave were sixteen round trips (16 % of a K = 384 launch).
  int ncol[NTW][4];
  bool nin[NTW][4];
  csg_f32x4 bv[NTW][4], gv[GATED ? 2 : 1][NTW][4];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ncol[nt][q] = wn * 32 * NTW + nt * 32 + 8 * q + 4 * h;
      nin[nt][q] = ncol[nt][q] < nrem;                // N % 4 == 0: a run is inside or outside as a whole
      bv[nt][q] = csg_f32x4{0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr && nin[nt][q]) bv[nt][q] = *(const csg_f32x4*)(bias + n0 + ncol[nt][q]);
    }
  if (GATED) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int m = wm * 64 + mt * 32 + c;
      const float* grow = gate + (int64_t)(m0 + m) * p.ldg + n0;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          gv[mt][nt][q] = csg_f32x4{1.f, 1.f, 1.f, 1.f};
          if (m < mrem && nin[nt][q]) gv[mt][nt][q] = *(const csg_f32x4*)(grow + ncol[nt][q]);
        }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v = acc[mt][nt][4 * q + u] + bv[nt][q][u];
          if (p.act == CSG_ACT_LEAKY) v = v > 0.f ? v : p.slope * v;
          if (GATED) v *= gv[GATED ? mt : 0][nt][q][u] > 0.f ? 1.f : p.gate_slope;
          acc[mt][nt][4 * q + u] = v;
        }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = wm * 64 + mt * 32 + c;
    const bool mok = m < mrem;
    float* yrow = Y + (int64_t)(m0 + m) * p.ldy + n0;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (mok && nin[nt][q])
          *(csg_f32x4*)(yrow + ncol[nt][q]) =
              csg_f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
  }
}

// ------------------------------------------------------------------------------------------------ dW = dY^T . X
// RS = rows (reduction steps) per stage: 32 (64 KB of LDS, two blocks per CU) or 16 (32 KB, four blocks per CU)
template <int RS>
__global__ __launch_bounds__(256, RS == 16 ? 4 : 2) void k_gemm_tn(GemmTnParams p, const float* __restrict__ dY,
                                                                   const float* __restrict__ X, float* __restrict__ out,
                                                                   float* __restrict__ dbout) {
  constexpr int ND = RS / 8;                          // DMA instructions per thread, operand and stage
  constexpr int HALF = RS * 128, STG = 2 * HALF;      // floats: one operand's tile, a stage
  constexpr int G = RS / 8;                           // groups of four k2-steps (eight rows) per stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 31, h = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  int bid = gm_xcd_remap(blockIdx.x, gridDim.x);
  // all output tiles of one row slice adjacent: the slice of dY and X stays in that XCD's L2
  const int kb = bid % p.nkb;
  bid /= p.nkb;
  const int nb = bid % p.nnb;
  const int sp = bid / p.nnb;
  const int n0 = nb * 128, k0 = kb * 128;
  const int r0 = sp * p.rows_per;
  const int rows = min(p.M - r0, p.rows_per);
  const int nstage = (rows + RS - 1) / RS;

  const float* Yb = dY + (int64_t)r0 * p.ldy + n0;
  const float* Xb = X + (int64_t)r0 * p.ldx + k0;
  const __amdgpu_buffer_rsrc_t rsY =
      __builtin_amdgcn_make_buffer_rsrc((void*)Yb, 0, (int)(((int64_t)(rows - 1) * p.ldy + (p.N - n0)) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)Xb, 0, (int)(((int64_t)(rows - 1) * p.ldx + (p.K - k0)) * 4), 0x00020000);

  // staging: DMA i fills LDS slot q = tid + 256 i of the [RS rows][32 slots] image: row tid / 32 + 8 i, columns 4 (tid % 32)
  const int col = (tid & 31) * 4, row0 = tid >> 5;
  const bool yok = col < p.N - n0, xok = col < p.K - k0;   // N, K multiples of 4
  unsigned offY[ND], offX[ND];
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    offY[i] = yok ? (unsigned)((row0 + 8 * i) * p.ldy + col) * 4u : CSG_OOB_OFF;
    offX[i] = xok ? (unsigned)((row0 + 8 * i) * p.ldx + col) * 4u : CSG_OOB_OFF;
  }
  auto dma_stage = [&](int s, int bufsel) {
    float* base = smem + bufsel * STG + wave * 256;
    const int left = rows - s * RS;                    // rows of this slice not yet staged
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const bool rin = row0 + 8 * i < left;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsY, (lds_ptr)(base + i * 1024), 16, (int)(rin ? offY[i] : CSG_OOB_OFF),
                                               s * RS * p.ldy * 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const bool rin = row0 + 8 * i < left;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(base + HALF + i * 1024), 16,
                                               (int)(rin ? offX[i] : CSG_OOB_OFF), s * RS * p.ldx * 4, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.f;
  const bool do_db = dbout != nullptr && kb == 0 && wn == 0;
  float dbacc[2] = {0.f, 0.f};

  // k2-step j of a stage: rows 2 j + h of the tiles; dY[.][64 wm + 32 t + c] and X[.][64 wn + 32 t + c]
  const float* ly = smem + h * 128 + wm * 64 + c;
  const float* lx = smem + HALF + h * 128 + wn * 64 + c;
  struct Frag {
    float e[4][2], v[4][2];
  };
  auto read_frag = [&](int bufsel, int j4, Frag& f) {   // four k2-steps: j = 4 j4 .. 4 j4 + 3
    const float* by = ly + bufsel * STG;
    const float* bx = lx + bufsel * STG;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 4 * j4 + u;
      f.e[u][0] = by[j * 256];
      f.e[u][1] = by[j * 256 + 32];
      f.v[u][0] = bx[j * 256];
      f.v[u][1] = bx[j * 256 + 32];
    }
  };
  auto compute = [&](const Frag& f) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.v[u][nt], f.e[u][mt], acc[mt][nt], 0, 0, 0);
      if (do_db) {
        dbacc[0] += f.e[u][0];
        dbacc[1] += f.e[u][1];
      }
    }
  };

  Frag f[2];
  if (nstage > 0) {
    dma_stage(0, 0);
    __syncthreads();
    read_frag(0, 0, f[0]);
    for (int s = 0; s < nstage; ++s) {
      const int buf = s & 1;
      if (s + 1 < nstage) dma_stage(s + 1, buf ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g + 1 < G; ++g) {
        read_frag(buf, g + 1, f[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        compute(f[g & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
      if (s + 1 < nstage) read_frag(buf ^ 1, 0, f[0]);
      compute(f[(G - 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: slab [sp][N][K] (or dw itself).  X is the first MFMA operand: D[i = k][j = n] — a lane holds one row
  // n = c and runs of four consecutive columns k: 16-byte stores
  float* dst = out + (p.direct ? 0 : (int64_t)sp * p.N * p.K);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int n = n0 + wm * 64 + mt * 32 + c;
    float* drow = dst + (int64_t)n * p.K;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = k0 + wn * 64 + nt * 32 + 8 * q + 4 * h;
        if (n < p.N && k < p.K) {
          const csg_f32x4 v = {acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]};
          *(csg_f32x4*)(drow + k) = v;
        }
      }
    }
  }
  if (do_db) {
    // the two lane halves hold the even / odd rows of every pair: one cross-half add, then 32 lanes write 32 columns
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float t = dbacc[mt] + __shfl_xor(dbacc[mt], 32);
      const int n = n0 + wm * 64 + mt * 32 + c;
      if (h == 0 && n < p.N) dbout[(p.direct ? 0 : (int64_t)sp * p.N) + n] = t;
    }
  }
}

static int64_t tn_plan(int64_t M, int64_t N, int64_t K, int* nsplit, int* rows_per) {
  const int64_t tiles = cdiv(N, 128) * cdiv(K, 128);
  // two blocks per CU: aim at >= 1024 blocks (two rounds of the 512 slots) while a slice keeps >= 256 rows
  int64_t want = cdiv(1024, tiles);
  const int64_t most = M / 256 > 0 ? M / 256 : 1;
  if (want > most) want = most;
  if (want < 1) want = 1;
  int64_t per = cdiv(cdiv(M, want), 32) * 32;
  const int64_t ns = cdiv(M, per);
  *nsplit = (int)ns;
  *rows_per = (int)per;
  return ns;
}

static bool nt_ok(const csg_gemm_desc* d) {
  return d->M > 0 && d->N > 0 && d->K > 0 && d->K % 4 == 0 && d->N % 4 == 0 && d->lda % 4 == 0 && d->ldb % 4 == 0 &&
         d->ldy % 4 == 0 && d->lda >= d->K &&
         d->ldb >= d->K && d->ldy >= d->N && (int64_t)128 * d->lda * 4 < (1ll << 30) && (int64_t)128 * d->ldb * 4 < (1ll << 30) &&
         d->M < (1ll << 31) - 128 && d->N < (1ll << 31) - 128 && cdiv(d->M, 128) * cdiv(d->N, 128) < (1ll << 31);
}

}  // namespace

extern "C" {

int csg_gemm_supported(const csg_gemm_desc* d) { return (d != nullptr && nt_ok(d)) ? 1 : 0; }

int csg_gemm_nt(const csg_gemm_desc* d, const float* a, const float* bw, const float* bias, const float* gate, float* y,
                void* stream) {
  CSG_REQUIRE(d != nullptr && a != nullptr && bw != nullptr && y != nullptr, CSG_E_BADSHAPE, "csg_gemm_nt: null pointer");
  CSG_REQUIRE(nt_ok(d), CSG_E_UNSUPPORTED,
              "csg_gemm_nt: needs K, N, lda, ldb, ldy multiples of 4, lda/ldb >= K, ldy >= N (M %lld N %lld K %lld lda %lld ldb %lld ldy %lld)",
              (long long)d->M, (long long)d->N, (long long)d->K, (long long)d->lda, (long long)d->ldb, (long long)d->ldy);
  CSG_REQUIRE((((uintptr_t)a | (uintptr_t)bw | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)gate) & 15) == 0, CSG_E_BADSHAPE,
              "csg_gemm_nt: pointers must be 16-byte aligned");
  CSG_REQUIRE(gate == nullptr || d->ldg % 4 == 0, CSG_E_BADSHAPE, "csg_gemm_nt: ldg must be a multiple of 4");
  CSG_REQUIRE(d->act == CSG_ACT_NONE || d->act == CSG_ACT_LEAKY, CSG_E_UNSUPPORTED,
              "csg_gemm_nt: activation %d", (int)d->act);
  CSG_REQUIRE(gate == nullptr || d->ldg >= d->N, CSG_E_BADSHAPE, "csg_gemm_nt: ldg < N");
  hipStream_t s = (hipStream_t)stream;
  GemmParams p;
  p.M = (int)d->M, p.N = (int)d->N, p.K = (int)d->K;
  p.lda = (int)d->lda, p.ldb = (int)d->ldb, p.ldy = (int)d->ldy, p.ldg = (int)d->ldg;
  p.act = d->act, p.slope = d->slope, p.gate_slope = d->gate_slope;
  const bool narrow = d->N <= 64;                      // 128 x 64 tiles
  p.nnb = (int)cdiv(d->N, narrow ? 64 : 128);
  p.nstage = (int)cdiv(d->K, 32);
  const unsigned grid = (unsigned)(cdiv(d->M, 128) * p.nnb);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_nt<2, 32, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)k_gemm_nt<2, 32, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)k_gemm_tn<32>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    attr = true;
  }
  ProfScope ps(K_GEMM_NT, 2.0 * (double)d->M * (double)d->N * (double)d->K, s);
  // (BK, NBUF) by CSG_GEMM_CFG = 322 | 162 — developer knob, see the kernel's header
  static const int cfg_env = getenv("CSG_GEMM_CFG") ? atoi(getenv("CSG_GEMM_CFG")) : 0;
  // measured (tools/gemm_probe.py, M = 96 000): four 32 KB blocks per CU beat two 64 KB ones by 2-4 % on 128-wide tiles; a
  // deeper ring — (16, 3) three blocks, (16, 4) two blocks — equals the two-buffer form at the same occupancy: the DMA latency
  // is not what is exposed.  The 64-wide tiles of N <= 64 prefer the long stages (84 vs 72 TFLOP/s at K = 128).
  const int cfg = cfg_env ? cfg_env : (narrow ? 322 : 162);
  // only these two (BK, NBUF) pairs are instantiated; any other value would size the LDS for a kernel that is not launched
  CSG_REQUIRE(cfg == 162 || cfg == 322, CSG_E_UNSUPPORTED, "csg_gemm_nt: CSG_GEMM_CFG=%d (only 162 and 322 are built)", cfg);
  const int bk = cfg / 10, nbuf = cfg % 10;
  p.nstage = (int)cdiv(d->K, bk);
  const size_t shm = (size_t)nbuf * (128 + (narrow ? 64 : 128)) * bk * 4;
#define GM_NT(NTW, BK, NBUF)                                                                                     \
  do {                                                                                                           \
    if (gate != nullptr)                                                                                         \
      CSG_LAUNCH((k_gemm_nt<NTW, BK, NBUF, true>), dim3(grid), dim3(256), shm, s, p, a, bw, bias, gate, y);      \
    else                                                                                                         \
      CSG_LAUNCH((k_gemm_nt<NTW, BK, NBUF, false>), dim3(grid), dim3(256), shm, s, p, a, bw, bias, gate, y);     \
  } while (0)
  if (cfg == 162) {
    if (narrow) GM_NT(1, 16, 2); else GM_NT(2, 16, 2);
  } else {
    if (narrow) GM_NT(1, 32, 2); else GM_NT(2, 32, 2);
  }
#undef GM_NT
  return check_launch("k_gemm_nt");
}

int64_t csg_gemm_tn_workspace(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || N % 4 || K % 4) return -1;
  int ns, per;
  tn_plan(M, N, K, &ns, &per);
  return ns > 1 ? (int64_t)ns * (N * K + N) * 4 : 0;
}

int csg_gemm_tn(int64_t M, int64_t N, int64_t K, const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw,
                float* db, float* workspace, int64_t workspace_bytes, void* stream) {
  CSG_REQUIRE(dy != nullptr && x != nullptr && dw != nullptr, CSG_E_BADSHAPE, "csg_gemm_tn: null pointer");
  CSG_REQUIRE(M > 0 && N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0 && ldy >= N && ldx >= K,
              CSG_E_UNSUPPORTED, "csg_gemm_tn: needs N, K, ldy, ldx multiples of 4 (M %lld N %lld K %lld)", (long long)M,
              (long long)N, (long long)K);
  CSG_REQUIRE(M < (1ll << 31) && N < (1ll << 31) - 128 && K < (1ll << 31) - 128, CSG_E_UNSUPPORTED,
              "csg_gemm_tn: extents beyond 32-bit indexing (M %lld N %lld K %lld)", (long long)M, (long long)N, (long long)K);
  CSG_REQUIRE((((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) & 15) == 0, CSG_E_BADSHAPE, "csg_gemm_tn: pointers must be 16-byte aligned");
  int ns, per;
  tn_plan(M, N, K, &ns, &per);
  CSG_REQUIRE((int64_t)per * (ldy > ldx ? ldy : ldx) * 4 < (1ll << 31) - 4096, CSG_E_UNSUPPORTED,
              "csg_gemm_tn: a row slice of %d rows x %lld floats exceeds the 2 GB window of a buffer descriptor", per,
              (long long)(ldy > ldx ? ldy : ldx));
  const int64_t need = ns > 1 ? (int64_t)ns * (N * K + N) * 4 : 0;
  CSG_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need), CSG_E_WORKSPACE,
              "csg_gemm_tn: workspace of %lld bytes needed", (long long)need);
  hipStream_t s = (hipStream_t)stream;
  GemmTnParams p;
  p.M = (int)M, p.N = (int)N, p.K = (int)K, p.ldy = (int)ldy, p.ldx = (int)ldx;
  p.nnb = (int)cdiv(N, 128), p.nkb = (int)cdiv(K, 128);
  p.nsplit = ns, p.rows_per = per, p.direct = ns == 1 ? 1 : 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)k_gemm_nt<2, 32, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)k_gemm_nt<2, 32, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)k_gemm_tn<32>, hipFuncAttributeMaxDynamicSharedMemorySize, GM_LDS_BYTES);
    attr = true;
  }
  float* slabs = ns > 1 ? workspace : dw;
  float* dbs = db == nullptr ? nullptr : (ns > 1 ? workspace + (int64_t)ns * N * K : db);
  ProfScope ps(K_GEMM_TN, 2.0 * (double)M * (double)N * (double)K, s);
  // rows per stage: 16 (four 32 KB blocks per CU) measured 2-4 % ahead of 32 (two 64 KB blocks); developer knob
  static const int rs = getenv("CSG_GEMM_TN_ROWS") ? atoi(getenv("CSG_GEMM_TN_ROWS")) : 16;
  CSG_REQUIRE(rs == 16 || rs == 32, CSG_E_UNSUPPORTED, "csg_gemm_tn: CSG_GEMM_TN_ROWS=%d (only 16 and 32 are built)", rs);
  if (rs == 16)
    CSG_LAUNCH(k_gemm_tn<16>, dim3((unsigned)(p.nnb * p.nkb * ns)), dim3(256), GM_LDS_BYTES / 2, s, p, dy, x, slabs, dbs);
  else
    CSG_LAUNCH(k_gemm_tn<32>, dim3((unsigned)(p.nnb * p.nkb * ns)), dim3(256), GM_LDS_BYTES, s, p, dy, x, slabs, dbs);
  if (ns > 1) launch_slab_reduce(slabs, N * K, dw, dbs, db != nullptr ? N : 0, db, ns, s);
  return check_launch("k_gemm_tn");
}

}  // extern "C"
