// K3/K8/K11 — implicit-GEMM convolution on the fp32 matrix cores of gfx950.
//
// Reference call sites: nn.Conv2d in spade/models/networks/generator.py:28,46,
// architecture.py:29-32, normalization.py:89-94, discriminator.py:175-187; nn.Linear in
// sg2im/layers.py:10.  The reference computes in fp32, and v_mfma_f32_32x32x2_f32 is an exact
// k-ordered fp32 FMA chain, so this path keeps the reference's precision (no TF32/bf16).
//
// GEMM view (forward and backward-data): rows m = output pixels, cols n = output channels,
// k = (tap, input channel).  Block tile 128 x BN x 32 (BN = 128 / 64 / 32), 4 waves.  Both
// operands are staged global -> registers -> LDS as [row][32 k] with rows padded to 36 floats:
// every fragment read is one conflict-free ds_read_b128 that feeds FOUR MFMAs (the k index inside
// a group of 8 is permuted identically for A and B).  LDS is double-buffered: one barrier per
// K-tile, the next tile's global loads are in flight during the 64 MFMAs of the current one.
//
// Loads go through buffer descriptors (SRDs): a lane whose tap falls outside the image, whose
// row is past M/N or whose k is past K simply gets an out-of-range offset and the hardware returns
// zeros — the loader is branch-free, so hipcc interleaves its ~50 VALU ops with the MFMAs
// (the first version's exec-masked loads left 30 % of the matrix pipe idle: profiles/archive/r01b).
//
// Small grids (few output tiles, long K — the low-resolution layers and their backward-data) are
// split along K into slabs that a second ordered pass sums and finishes (bias/activation/residual);
// no atomics, results are reproducible.
//
// Weight gradient: rows = output channels (A = dY), cols = (tap, input channel) (B = gathered X),
// reduction over pixels, split across blocks into slabs + ordered reduction as well.
#include <stddef.h>
#include <stdlib.h>

#include <type_traits>

#include "csg_common.h"
#include "csg_reduce.h"

using namespace csg;

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 128-bit raw buffer load.  hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword
// load (llvm.amdgcn.raw.ptr.buffer.load.i32; verified on hardware with tools/probe), so the LLVM
// intrinsic is declared directly, with the descriptor as four plain dwords.
__device__ f32x4 csg_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");

__device__ __forceinline__ i32x4 make_srd(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = (int)(a & 0xffffffffu);
  r.y = (int)((a >> 32) & 0xffffu);   // stride 0: raw buffer, offsets in bytes
  r.z = (int)bytes;                   // num_records: loads at or past it return 0
  r.w = 0x00020000;                   // DATA_FORMAT_32
  return r;
}

#define IG_BM 128
#define IG_BK 32
#define IG_LD 36
#define OOB_OFF 0x80000000u           // >= any num_records we ever set: the load returns 0
#define MAX_RECORDS 0x7FFFFFF0ll

struct FastDiv {                      // exact n / d for 32-bit unsigned n (Granlund-Montgomery round-up)
  unsigned m, s1, s2;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = l < 1 ? l : 1;
  f.s2 = l > 0 ? l - 1 : 0;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  unsigned t = __umulhi(f.m, n);
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct IgemmParams {
  csg_conv_desc d;
  int M, Ktot, wrow;
  int mtiles, ntiles;
  int simple_out;  // output pixel index == m
  int ksplit, kt_per_split;
  FastDiv div_ow, div_oh;
  unsigned img_bytes;   // bytes of one input image (IHp*IWp*x_cs*4)
  long long ws_off;     // floats: where this launch's split-K slabs start in the workspace
  // tail split (ksplit == 1, tail_ks > 1): tiles [0, tail_tile0) run unsplit — whole rounds of the 512 resident
  // blocks — and only the m-tiles of the last, partly filled round are cut along K into tail_ks slabs
  int tail_tile0, tail_ks, tail_kt_per, tail_m0;
};

// Up to four descriptors served by ONE launch (blockIdx.y picks one): the parity classes of a stride-2 transposed
// convolution (PatchGAN backward-data).  Each class alone is a quarter of the pixels — grids of 35-140 tiles for 512
// block slots — and used to be its own launch plus its own split-K epilogue.
#define IG_MAXCLS 4
struct IgemmMulti {
  IgemmParams p[IG_MAXCLS];
};

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of tiles
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// The three tap tables are 48 consecutive ints of the by-value kernel argument.  Lanes 0..47 copy
// them from the kernarg segment with ONE vector load each (indexing them per lane through scalar
// registers would cost 48 SGPRs and spill the hot loop's scalars).
__device__ __forceinline__ void load_taps(int* s_tap, int tid, int karg_word = 0) {
  const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
  if (tid < 3 * CSG_MAX_TAPS) s_tap[tid] = ka[karg_word + offsetof(csg_conv_desc, tap_dy) / 4 + tid];
}

__device__ __forceinline__ void decompose(const IgemmParams& p, unsigned m, int& b, int& gy, int& gx) {
  unsigned t = fdiv(m, p.div_ow);
  gx = (int)(m - t * (unsigned)p.d.OWg);
  unsigned bb = fdiv(t, p.div_oh);
  gy = (int)(t - bb * (unsigned)p.d.OHg);
  b = (int)bb;
}


template <int BN, bool MULTI = false>
__global__ __launch_bounds__(256, 2) void k_igemm_fwd(typename std::conditional<MULTI, IgemmMulti, IgemmParams>::type karg,
                                                       const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ res,
                                                       float* __restrict__ y, float* __restrict__ ws) {
  const IgemmParams& p = *((const IgemmParams*)&karg + (MULTI ? blockIdx.y : 0));
  if (MULTI && (int)blockIdx.x >= p.mtiles * p.ntiles * p.ksplit) return;
  ws += p.ws_off;
  constexpr int MI = BN == 32 ? 1 : 2;       // 32x32 tiles per wave along m
  constexpr int NI = BN == 128 ? 2 : 1;      // ... along n
  constexpr int WM = BN == 32 ? 4 : 2;       // waves along m
  constexpr int BROWS = BN / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * IG_BM * IG_LD;
  int* s_tap = (int*)(Bs + 2 * BN * IG_LD);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const csg_conv_desc& d = p.d;
  int bid, ks = 0;
  int kt_per = p.kt_per_split;     // k tiles of this block's range
  bool to_slab = p.ksplit > 1;
  long long slab_rows = p.M;       // rows per slab and first row held by the slabs
  int slab_m0 = 0;
  if (p.tail_ks > 1) {
    if ((int)blockIdx.x < p.tail_tile0) {
      bid = xcd_remap(blockIdx.x, p.tail_tile0);
    } else {
      const int t = (int)blockIdx.x - p.tail_tile0;
      ks = t % p.tail_ks;
      bid = p.tail_tile0 + t / p.tail_ks;
      kt_per = p.tail_kt_per;
      to_slab = true;
      slab_m0 = p.tail_m0;
      slab_rows = p.M - p.tail_m0;
    }
  } else {
    bid = xcd_remap(blockIdx.x, p.mtiles * p.ntiles * p.ksplit);
    if (p.ksplit > 1) {
      ks = bid % p.ksplit;
      bid /= p.ksplit;
    }
  }
  const int mt = bid / p.ntiles, nt = bid - mt * p.ntiles;

  load_taps(s_tap, tid, MULTI ? (int)(blockIdx.y * (sizeof(IgemmParams) / 4)) : 0);

  // ---- buffer descriptors (wave-uniform): A based at the image of this tile's first row
  const int m_first = min(mt * IG_BM, p.M - 1);
  const int b0 = (int)fdiv(fdiv((unsigned)m_first, p.div_ow), p.div_oh);
  const long long a_rem = (long long)(d.B - b0) * p.img_bytes;
  const i32x4 rsA = make_srd(x + (long long)b0 * (p.img_bytes >> 2), (unsigned)(a_rem < MAX_RECORDS ? a_rem : MAX_RECORDS));
  const long long w_bytes = (long long)d.Cout * p.wrow * 4;
  const i32x4 rsB = make_srd(w, (unsigned)(w_bytes < MAX_RECORDS ? w_bytes : MAX_RECORDS));

  const int r0 = tid >> 3, kc = tid & 7;
  unsigned a_base[4];
  int a_iy0[4], a_ix0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = mt * IG_BM + r0 + 32 * i;
    const bool ok = m < p.M;
    int b, gy, gx;
    decompose(p, (unsigned)(ok ? m : 0), b, gy, gx);
    a_base[i] = (unsigned)(b - b0) * p.img_bytes;
    a_iy0[i] = ok ? gy * d.istride : -(1 << 28);
    a_ix0[i] = gx * d.istride;
  }
  unsigned b_base[BROWS];
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int n = nt * BN + r0 + 32 * i;
    b_base[i] = n < d.Cout ? (unsigned)n * (unsigned)p.wrow * 4u : OOB_OFF;
  }

  const int nkt_all = (p.Ktot + IG_BK - 1) / IG_BK;
  const int kt0 = ks * kt_per;
  const int kt1 = min(nkt_all, kt0 + kt_per);
  // this thread's k position (tap slot, channel) of the tile being loaded, advanced incrementally
  int k_cur = kt0 * IG_BK + kc * 4;
  int slot = k_cur / d.Cin;
  int cch = k_cur - slot * d.Cin;
  const int step_slots = IG_BK / d.Cin, step_rem = IG_BK - step_slots * d.Cin;

  f32x4 ra0[4], rb0[BROWS], ra1[4], rb1[BROWS];   // two sets: loads run two K tiles ahead of the MFMAs
  __syncthreads();  // s_tap visible

  // tap of the tile to be loaded next, fetched from LDS one tile ahead so that its latency hides
  // behind the MFMAs instead of heading the loader
  int t_dy, t_dx, t_tw;
  auto fetch_tap = [&]() {
    const int sl = (k_cur < p.Ktot) ? slot : 0;
    t_dy = s_tap[sl];
    t_dx = s_tap[16 + sl];
    t_tw = s_tap[32 + sl];
  };
  fetch_tap();

  auto load_tile = [&](f32x4* ra, f32x4* rb) {
    const bool kv = k_cur < p.Ktot;
    const int dy = t_dy, dx = t_dx, tw = t_tw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = a_iy0[i] + dy, ix = a_ix0[i] + dx;
      const bool inb = kv & ((unsigned)iy < (unsigned)d.IHv) & ((unsigned)ix < (unsigned)d.IWv);
      const unsigned off =
          a_base[i] + (unsigned)(((iy >> d.in_up) * d.IWp + (ix >> d.in_up)) * d.x_cs + cch) * 4u;
      ra[i] = csg_buffer_load_f32x4(rsA, (int)(inb ? off : OOB_OFF), 0, 0);
    }
    const unsigned wcol = (unsigned)(tw * d.Cin + cch) * 4u;
#pragma unroll
    for (int i = 0; i < BROWS; ++i)
      rb[i] = csg_buffer_load_f32x4(rsB, (int)((kv & (b_base[i] != OOB_OFF)) ? b_base[i] + wcol : OOB_OFF), 0, 0);
    // advance to the next K tile (branch-free: 32 = step_slots * Cin + step_rem)
    k_cur += IG_BK;
    cch += step_rem;
    slot += step_slots;
    const bool wrap = cch >= d.Cin;
    cch = wrap ? cch - d.Cin : cch;
    slot = wrap ? slot + 1 : slot;
    fetch_tap();
  };
  auto store_tile = [&](int buf, const f32x4* ra, const f32x4* rb) {
    float* a = As + buf * IG_BM * IG_LD + r0 * IG_LD + kc * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(a + 32 * i * IG_LD) = ra[i];
    float* b = Bs + buf * BN * IG_LD + r0 * IG_LD + kc * 4;
#pragma unroll
    for (int i = 0; i < BROWS; ++i) *(f32x4*)(b + 32 * i * IG_LD) = rb[i];
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int wm = BN == 32 ? wave : (wave >> 1), wn = BN == 32 ? 0 : (wave & 1);
  const int r = lane & 31, hh = lane >> 5;
  constexpr int WROWS = IG_BM / WM;      // A rows per wave
  constexpr int WCOLS = BN / (4 / WM);   // B rows (output channels) per wave

  if (kt0 < kt1) {
    load_tile(ra0, rb0);
    store_tile(0, ra0, rb0);
  }
  if (kt0 + 1 < kt1) load_tile(ra1, rb1);
  __syncthreads();

  auto compute_tile = [&](int buf) {
    const float* Ab = As + buf * IG_BM * IG_LD + (wm * WROWS + r) * IG_LD + 4 * hh;
    const float* Bb = Bs + buf * BN * IG_LD + (wn * WCOLS + r) * IG_LD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 a[MI], b[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[mi] = *(const float4*)(Ab + mi * 32 * IG_LD + g * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[ni] = *(const float4*)(Bb + ni * 32 * IG_LD + g * 8);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          // weights as the first operand: D[i = channel][j = pixel], so each lane holds one pixel
          // and runs of 4 consecutive channels -> 16-byte epilogue stores
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].x, a[mi].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].y, a[mi].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].z, a[mi].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ni].w, a[mi].w, acc[mi][ni], 0, 0, 0);
        }
    }
  };
  // MFMAs of tile `buf` with the LDS refill of the other buffer spread between the four k-groups, so that
  // the ds_writes drain under the MFMAs instead of queueing up in front of the barrier
  auto compute_store = [&](int buf, const f32x4* ra, const f32x4* rb) {
    const float* Ab = As + buf * IG_BM * IG_LD + (wm * WROWS + r) * IG_LD + 4 * hh;
    const float* Bb = Bs + buf * BN * IG_LD + (wn * WCOLS + r) * IG_LD + 4 * hh;
    float* sa = As + (buf ^ 1) * IG_BM * IG_LD + r0 * IG_LD + kc * 4;
    float* sb = Bs + (buf ^ 1) * BN * IG_LD + r0 * IG_LD + kc * 4;
    // fragments of k-group g+1 are fetched before the MFMAs of group g are issued
    float4 a[2][MI], b[2][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) a[0][mi] = *(const float4*)(Ab + mi * 32 * IG_LD);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) b[0][ni] = *(const float4*)(Bb + ni * 32 * IG_LD);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (g + 1 < 4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[(g + 1) & 1][mi] = *(const float4*)(Ab + mi * 32 * IG_LD + (g + 1) * 8);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[(g + 1) & 1][ni] = *(const float4*)(Bb + ni * 32 * IG_LD + (g + 1) * 8);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[g & 1][ni].x, a[g & 1][mi].x, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[g & 1][ni].y, a[g & 1][mi].y, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[g & 1][ni].z, a[g & 1][mi].z, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[g & 1][ni].w, a[g & 1][mi].w, acc[mi][ni], 0, 0, 0);
        }
      *(f32x4*)(sa + 32 * g * IG_LD) = ra[g];
      if (g < BROWS) *(f32x4*)(sb + 32 * g * IG_LD) = rb[g];
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // steady state: branch-free body, two K tiles per trip.  Invariant at the top: tile kt is in
  // LDS[buf] and tile kt+1 is in flight in register set 1; the loads issued in a half-trip are only
  // consumed (LDS refill) one half-trip later, so their latency hides behind 64 MFMAs.
  int kt = kt0, buf = 0;
  for (; kt + 3 < kt1; kt += 2) {
    load_tile(ra0, rb0);
    compute_store(buf, ra1, rb1);
    __syncthreads();
    load_tile(ra1, rb1);
    compute_store(buf ^ 1, ra0, rb0);
    __syncthreads();
  }
  const int left = kt1 - kt;            // 0..3 tiles remain
  if (left == 3) {
    load_tile(ra0, rb0);
    compute_store(buf, ra1, rb1);
    __syncthreads();
    compute_store(buf ^ 1, ra0, rb0);
    __syncthreads();
    compute_tile(buf);
  } else if (left == 2) {
    compute_store(buf, ra1, rb1);
    __syncthreads();
    compute_tile(buf ^ 1);
  } else if (left == 1) {
    compute_tile(buf);
  }

  // epilogue: D[i][j], j = lane&31 = pixel, i = (reg&3) + 8*(reg>>2) + 4*(lane>>5) = channel:
  // registers 4g..4g+3 of a tile are channels 8g + 4*(lane>>5) + 0..3 of this lane's pixel.
  const bool vec = ((d.Cout | d.y_cs) & 3) == 0;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = mt * IG_BM + wm * WROWS + mi * 32 + r;
    if (m >= p.M) continue;
    if (to_slab) {  // raw partial sums into this split's slab (Cout % 4 == 0 on this path)
      float* srow = ws + ((long long)ks * slab_rows + (m - slab_m0)) * d.Cout;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = nt * BN + wn * WCOLS + ni * 32 + 8 * g + 4 * hh;
          if (n < d.Cout)
            *(float4*)(srow + n) = make_float4(acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2],
                                               acc[mi][ni][4 * g + 3]);
        }
      continue;
    }
    long long pix = m;
    if (!p.simple_out) {
      int b, gy, gx;
      decompose(p, (unsigned)m, b, gy, gx);
      pix = ((long long)b * d.OHf + gy * d.os + d.ooy) * d.OWf + gx * d.os + d.oox;
    }
    float* yrow = y + pix * d.y_cs;
    const float* rrow = res != nullptr ? res + pix * d.y_cs : nullptr;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nt * BN + wn * WCOLS + ni * 32 + 8 * g + 4 * hh;
        if (n >= d.Cout) continue;
        float v[4] = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
        if (vec) {
          if (bias != nullptr) {
            const float4 bv = *(const float4*)(bias + n);
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (d.act == CSG_ACT_LEAKY)
              v[j] = v[j] > 0.f ? v[j] : v[j] * d.slope;
            else if (d.act == CSG_ACT_TANH)
              v[j] = tanhf(v[j]);
          }
          if (rrow != nullptr) {
            const float4 rv = *(const float4*)(rrow + n);
            if (d.res_gate) {
              v[0] *= rv.x > 0.f ? 1.f : d.slope; v[1] *= rv.y > 0.f ? 1.f : d.slope;
              v[2] *= rv.z > 0.f ? 1.f : d.slope; v[3] *= rv.w > 0.f ? 1.f : d.slope;
            } else {
              v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
            }
          }
          if (d.accumulate) {
            const float4 ov = *(const float4*)(yrow + n);
            v[0] += ov.x; v[1] += ov.y; v[2] += ov.z; v[3] += ov.w;
          }
          *(float4*)(yrow + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (n + j < d.Cout) {
              float t = v[j] + (bias != nullptr ? bias[n + j] : 0.f);
              if (d.act == CSG_ACT_LEAKY)
                t = t > 0.f ? t : t * d.slope;
              else if (d.act == CSG_ACT_TANH)
                t = tanhf(t);
              if (rrow != nullptr) t = d.res_gate ? t * (rrow[n + j] > 0.f ? 1.f : d.slope) : t + rrow[n + j];
              if (d.accumulate) t += yrow[n + j];
              yrow[n + j] = t;
            }
          }
        }
      }
  }
}

// second pass of a split-K launch: ordered slab sum + the forward epilogue, 4 channels per thread
template <bool MULTI = false>
__global__ __launch_bounds__(256) void k_splitk_epilogue(typename std::conditional<MULTI, IgemmMulti, IgemmParams>::type karg,
                                                          const float* __restrict__ ws, const float* __restrict__ bias,
                                                          const float* __restrict__ res, float* __restrict__ y) {
  const IgemmParams& p = *((const IgemmParams*)&karg + (MULTI ? blockIdx.y : 0));
  if (MULTI && p.ksplit <= 1) return;
  ws += p.ws_off;
  const csg_conv_desc& d = p.d;
  const int q = d.Cout >> 2;
  // tail split: the slabs hold rows [tail_m0, M) only
  const int m0 = p.tail_ks > 1 ? p.tail_m0 : 0;
  const int nsl = p.tail_ks > 1 ? p.tail_ks : p.ksplit;
  const long long n4 = (long long)(p.M - m0) * q;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (long long)gridDim.x * blockDim.x) {
    const unsigned ml = (unsigned)(e / q);
    const unsigned m = ml + (unsigned)m0;
    const int n = (int)(e - (long long)ml * q) * 4;
    const float* src = ws + (long long)ml * d.Cout + n;
    float4 a = *(const float4*)src;
    const long long slab = (long long)(p.M - m0) * d.Cout;
    int s = 1;
    for (; s + 4 <= nsl; s += 4) {          // four loads in flight, added in slab order
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *(const float4*)(src + (s + u) * slab);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
    }
    for (; s < nsl; ++s) {
      const float4 t = *(const float4*)(src + s * slab);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    long long pix = m;
    if (!p.simple_out) {
      int b, gy, gx;
      decompose(p, m, b, gy, gx);
      pix = ((long long)b * d.OHf + gy * d.os + d.ooy) * d.OWf + gx * d.os + d.oox;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    float* dst = y + pix * d.y_cs + n;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t = v[j] + (bias != nullptr ? bias[n + j] : 0.f);
      if (d.act == CSG_ACT_LEAKY)
        t = t > 0.f ? t : t * d.slope;
      else if (d.act == CSG_ACT_TANH)
        t = tanhf(t);
      if (res != nullptr) {
        const float rv = res[pix * d.y_cs + n + j];
        t = d.res_gate ? t * (rv > 0.f ? 1.f : d.slope) : t + rv;
      }
      if (d.accumulate) t += dst[j];
      dst[j] = t;
    }
  }
}

// ------------------------------------------------------------------------------- weight grad
// tile: BI output channels (i) x 128 (tap,cin) columns (j), 32 pixels per reduction step
#define WG_LDB 128
template <int BI>
__global__ __launch_bounds__(256, 2) void k_igemm_wgrad(IgemmParams p, const float* __restrict__ x,
                                                         const float* __restrict__ dy, float* __restrict__ out,
                                                         int itiles, int jtiles, int nsplit, int cps,
                                                         float* __restrict__ dbp) {
  constexpr int WI = BI == 32 ? 1 : 2;          // waves along i
  constexpr int WJ = 4 / WI;                    // waves along j
  constexpr int MI = BI / (WI * 32);            // 32x32 tiles per wave along i
  constexpr int NJ = 128 / (WJ * 32);           // ... along j
  constexpr int CQ = BI / 4;                    // float4 columns of the dY tile
  constexpr int RP = 256 / CQ;                  // dY rows loaded per pass
  constexpr int AIT = 32 / RP;                  // passes
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                      // [2][32][BI]   dY tile, [pixel][n]
  float* Bs = smem + 2 * 32 * BI;        // [2][32][128]  X tile,  [pixel][kk]
  int* s_tap = (int*)(Bs + 2 * 32 * WG_LDB);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const csg_conv_desc& d = p.d;
  // block order: all (channel tile, tap tile) pairs of one pixel split are adjacent and land on the
  // same XCD, so the split's dY rows and X pixels are fetched from HBM once and re-read from that L2
  int bid = xcd_remap(blockIdx.x, itiles * jtiles * nsplit);
  const int sp = bid / (itiles * jtiles);
  bid -= sp * (itiles * jtiles);
  const int jt = bid % jtiles, it = bid / jtiles;

  load_taps(s_tap, tid);
  __syncthreads();

  // X tile: this thread's column is a fixed (tap, channel)
  const int pr = tid >> 5, c4 = tid & 31;
  const int kk0 = jt * 128 + c4 * 4;
  const bool kv = kk0 < p.Ktot;
  const int slot = kv ? kk0 / d.Cin : 0;
  const int cch = kk0 - slot * d.Cin;
  const int tdy = s_tap[slot], tdx = s_tap[16 + slot];
  // dY tile: this thread's column
  const int apr = tid / CQ, ac4 = tid % CQ;
  const int n0 = it * BI + ac4 * 4;
  const unsigned a_col = n0 < d.Cout ? (unsigned)n0 * 4u : OOB_OFF;   // Cout % 4 == 0

  const int nch = (p.M + 31) / 32;
  const int ch0 = sp * cps, ch1 = min(nch, ch0 + cps);

  // X descriptor: based at the image of the split's first pixel (a split may span many images only
  // when images are small, so offsets stay far below 2 GB; checked on the host)
  const int m_first = min(ch0 * 32, p.M - 1);
  const int b0 = (int)fdiv(fdiv((unsigned)m_first, p.div_ow), p.div_oh);
  const long long x_rem = (long long)(d.B - b0) * p.img_bytes;
  const i32x4 rsX = make_srd(x + (long long)b0 * (p.img_bytes >> 2), (unsigned)(x_rem < MAX_RECORDS ? x_rem : MAX_RECORDS));

  // two register sets: the loads of chunk t+2 are issued during the MFMAs of chunk t and consumed
  // (LDS refill) at the end of chunk t+1, so their latency has a whole chunk to hide behind
  f32x4 ra0[AIT], rb0[4], ra1[AIT], rb1[4];
  auto load_tile = [&](int ch, f32x4* ra, f32x4* rb) {
    // dY rows of this chunk: descriptor re-based per chunk, rows past M fall out of range
    const int rows = min(32, p.M - ch * 32);
    const i32x4 rsY = make_srd(dy + (long long)ch * 32 * d.y_cs, (unsigned)(rows * d.y_cs * 4));
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const unsigned off = (unsigned)((apr + RP * i) * d.y_cs) * 4u;
      ra[i] = csg_buffer_load_f32x4(rsY, (int)(a_col != OOB_OFF ? off + a_col : OOB_OFF), 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = ch * 32 + pr + 8 * i;
      const bool ok = m < p.M;
      int b, gy, gx;
      decompose(p, (unsigned)(ok ? m : 0), b, gy, gx);
      const int iy = gy * d.istride + tdy, ix = gx * d.istride + tdx;
      const bool inb = ok & kv & ((unsigned)iy < (unsigned)d.IHv) & ((unsigned)ix < (unsigned)d.IWv);
      const unsigned off = (unsigned)(b - b0) * p.img_bytes +
                           (unsigned)(((iy >> d.in_up) * d.IWp + (ix >> d.in_up)) * d.x_cs + cch) * 4u;
      rb[i] = csg_buffer_load_f32x4(rsX, (int)(inb ? off : OOB_OFF), 0, 0);
    }
  };
  // bias gradient (column sums of dY) for free: the j-tile-0 block of every (channel tile, split) adds up
  // the dY values it stages anyway (rows past M arrive as zeros)
  const bool do_db = dbp != nullptr && jt == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  auto store_tile = [&](int buf, const f32x4* ra, const f32x4* rb) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) *(f32x4*)(As + buf * 32 * BI + (apr + RP * i) * BI + ac4 * 4) = ra[i];
    if (do_db) {
#pragma unroll
      for (int i = 0; i < AIT; ++i) csum += ra[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *(f32x4*)(Bs + buf * 32 * WG_LDB + (pr + 8 * i) * WG_LDB + c4 * 4) = rb[i];
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int wi = BI == 32 ? 0 : (wave >> 1), wj = BI == 32 ? wave : (wave & 1);
  const int r = lane & 31, hh = lane >> 5;

  // Fragment rows are INTERLEAVED: lane r of tile mi holds channel wi*(MI*32) + r*MI + mi (and column
  // r*NJ + nj of the X tile), so one ds_read_b64 fetches a lane's operands for MI (NJ) tiles at once —
  // half the LDS instructions of one ds_read_b32 per tile.  The epilogue undoes the permutation.
  auto frag = [&](const float* ptr, float* dst, auto n_tag) {
    constexpr int N = decltype(n_tag)::value;
    if constexpr (N == 2) {
      const float2 t = *(const float2*)ptr;
      dst[0] = t.x;
      dst[1] = t.y;
    } else {
      dst[0] = *ptr;
    }
  };
  auto compute_tile = [&](int buf) {
    const float* Ab = As + buf * 32 * BI + hh * BI + wi * (MI * 32) + r * MI;
    const float* Bb = Bs + buf * 32 * WG_LDB + hh * WG_LDB + wj * (NJ * 32) + r * NJ;
    // fragments of pixel pair kp+1 are fetched from LDS before the MFMAs of pair kp are issued
    float a[2][MI], b[2][NJ];
    frag(Ab, a[0], std::integral_constant<int, MI>());
    frag(Bb, b[0], std::integral_constant<int, NJ>());
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      if (kp + 1 < 16) {
        frag(Ab + (kp + 1) * 2 * BI, a[(kp + 1) & 1], std::integral_constant<int, MI>());
        frag(Bb + (kp + 1) * 2 * WG_LDB, b[(kp + 1) & 1], std::integral_constant<int, NJ>());
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[kp & 1][nj], a[kp & 1][mi], acc[mi][nj], 0, 0, 0);
    }
  };

  // the same MFMAs with the LDS refill of the other buffer (AIT dY pieces + 4 X pieces) spread over the 16 pixel
  // pairs: the ds_writes drain under the MFMAs instead of queueing up in front of the barrier
  auto compute_store = [&](int buf, const f32x4* ra, const f32x4* rb) {
    const float* Ab = As + buf * 32 * BI + hh * BI + wi * (MI * 32) + r * MI;
    const float* Bb = Bs + buf * 32 * WG_LDB + hh * WG_LDB + wj * (NJ * 32) + r * NJ;
    float* sa = As + (buf ^ 1) * 32 * BI + apr * BI + ac4 * 4;
    float* sb = Bs + (buf ^ 1) * 32 * WG_LDB + pr * WG_LDB + c4 * 4;
    float a[2][MI], b[2][NJ];
    frag(Ab, a[0], std::integral_constant<int, MI>());
    frag(Bb, b[0], std::integral_constant<int, NJ>());
#pragma unroll
    for (int kp = 0; kp < 16; ++kp) {
      if (kp + 1 < 16) {
        frag(Ab + (kp + 1) * 2 * BI, a[(kp + 1) & 1], std::integral_constant<int, MI>());
        frag(Bb + (kp + 1) * 2 * WG_LDB, b[(kp + 1) & 1], std::integral_constant<int, NJ>());
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[kp & 1][nj], a[kp & 1][mi], acc[mi][nj], 0, 0, 0);
      if ((kp & 1) == 1) {                       // 8 slots: kp = 1, 3, ..., 15
        const int slot = kp >> 1;
        if (slot < AIT) {
          *(f32x4*)(sa + RP * slot * BI) = ra[slot];
          if (do_db) csum += ra[slot];
        }
        if (slot >= 4) *(f32x4*)(sb + 8 * (slot - 4) * WG_LDB) = rb[slot - 4];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    static_assert(AIT <= 4, "the refill slots cover at most 4 dY pieces");
  };

  // invariant at the top of the steady loop: chunk `ch` is in LDS[buf], chunk ch+1 is in flight in set 1
  int ch = ch0, buf = 0;
  if (ch0 < ch1) {
    load_tile(ch0, ra0, rb0);
    store_tile(0, ra0, rb0);
  }
  if (ch0 + 1 < ch1) load_tile(ch0 + 1, ra1, rb1);
  __syncthreads();
  for (; ch + 3 < ch1; ch += 2) {      // branch-free body: two chunks per trip
    load_tile(ch + 2, ra0, rb0);
    compute_store(buf, ra1, rb1);
    __syncthreads();
    load_tile(ch + 3, ra1, rb1);
    compute_store(buf ^ 1, ra0, rb0);
    __syncthreads();
  }
  const int left = ch1 - ch;            // 0..3 chunks remain
  if (left == 3) {
    load_tile(ch + 2, ra0, rb0);
    compute_store(buf, ra1, rb1);
    __syncthreads();
    compute_store(buf ^ 1, ra0, rb0);
    __syncthreads();
    compute_tile(buf);
  } else if (left == 2) {
    compute_store(buf, ra1, rb1);
    __syncthreads();
    compute_tile(buf ^ 1);
  } else if (left == 1) {
    compute_tile(buf);
  }

  // D[i = kk][j = n]: a lane owns output channel n and, per (g, hh), a run of 4*NJ consecutive (tap,c)
  // columns (row rho = 8g+4hh+q of tile nj is column rho*NJ + nj), contiguous in the slab row
  // [split][Cout][wrow] (Cin % 4 == 0, so a float4 never straddles taps)
  float* slab = out + (long long)sp * d.Cout * p.wrow;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int n = it * BI + wi * (MI * 32) + r * MI + mi;
    if (n >= d.Cout) continue;
    float* srow = slab + (long long)n * p.wrow;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int f = 0; f < NJ; ++f) {
        const int kk = jt * 128 + wj * (NJ * 32) + (8 * g + 4 * hh) * NJ + 4 * f;
        if (kk < p.Ktot) {
          const int sl = kk / d.Cin;
          const int col = s_tap[32 + sl] * d.Cin + (kk - sl * d.Cin);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int L = 4 * f + e;
            v[e] = acc[mi][L % NJ][4 * g + L / NJ];
          }
          *(float4*)(srow + col) = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
  }
  if (do_db) {                               // uniform per block
    __syncthreads();                         // every wave is done with the LDS tiles
    f32x4* red = (f32x4*)smem;               // [RP][CQ]
    red[apr * CQ + ac4] = csum;
    __syncthreads();
    if (apr == 0 && n0 < d.Cout) {
      f32x4 tot = red[ac4];
      for (int q = 1; q < RP; ++q) tot += red[q * CQ + ac4];      // fixed order
      *(f32x4*)(dbp + (long long)sp * d.Cout + n0) = tot;
    }
  }
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
  // 32-bit buffer offsets: a 128-row tile may straddle images; keep that span below 2 GB
  const int64_t img_bytes = (int64_t)d->IHp * d->IWp * d->x_cs * 4;
  const int64_t npix = (int64_t)d->OHg * d->OWg;
  const int64_t span = (127 / npix + 2) * img_bytes;
  CSG_REQUIRE(span < MAX_RECORDS, CSG_E_UNSUPPORTED, "%s: one input image is too large for 32-bit offsets", who);
  CSG_REQUIRE((int64_t)d->Cout * d->wtaps * d->Cin * 4 < MAX_RECORDS, CSG_E_UNSUPPORTED, "%s: weights too large", who);
  return CSG_OK;
}

static void fill(IgemmParams& p, const csg_conv_desc* d) {
  p.d = *d;
  p.M = d->B * d->OHg * d->OWg;
  p.Ktot = d->ntaps * d->Cin;
  p.wrow = d->wtaps * d->Cin;
  p.simple_out = (d->os == 1 && d->ooy == 0 && d->oox == 0 && d->OHg == d->OHf && d->OWg == d->OWf) ? 1 : 0;
  p.ksplit = 1;
  p.kt_per_split = (p.Ktot + IG_BK - 1) / IG_BK;
  p.div_ow = make_fastdiv((unsigned)d->OWg);
  p.div_oh = make_fastdiv((unsigned)d->OHg);
  p.img_bytes = (unsigned)((int64_t)d->IHp * d->IWp * d->x_cs * 4);
  p.mtiles = p.ntiles = 0;
  p.ws_off = 0;
  p.tail_tile0 = p.tail_ks = p.tail_kt_per = p.tail_m0 = 0;
}

static int pick_bn(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

// Number of K (or pixel) splits for a grid of `blocks` tiles with `steps` reduction steps each.
// 512 blocks are resident at once (2 per CU); the launch takes ceil(blocks*s/512) rounds of
// steps/s each.  Pick the smallest s whose cost is within 3 % of the best (slabs cost traffic).
static int pick_split(int blocks, int steps, int min_steps, int max_split) {
  int best = 1;
  double best_cost = 1e30;
  for (int s = 1; s <= max_split && steps / s >= min_steps; ++s) {
    const double rounds = (double)((blocks * (long long)s + 511) / 512);
    const double cost = rounds * ((steps + s - 1) / s) + 0.25 * s;   // + slab write/reduce per split
    if (cost < best_cost * 0.97) {
      best_cost = cost;
      best = s;
    }
  }
  return best;
}

// split-K plan for the forward kernel: only when the tile grid cannot fill the chip
static void fwd_plan(IgemmParams& p) {
  const int bn = pick_bn(p.d.Cout);
  p.mtiles = (p.M + IG_BM - 1) / IG_BM;
  p.ntiles = (p.d.Cout + bn - 1) / bn;
  const int nkt = (p.Ktot + IG_BK - 1) / IG_BK;
  const int blocks = p.mtiles * p.ntiles;
  int ks = 1;
  if (blocks < 1024 && nkt >= 16 && p.d.Cout % 4 == 0) ks = pick_split(blocks, nkt, 8, 64);
  static const int force = getenv("CSG_IGEMM_KSPLIT") ? atoi(getenv("CSG_IGEMM_KSPLIT")) : 0;   // experiments only
  if (force > 0 && p.d.Cout % 4 == 0) ks = force < nkt ? force : nkt;
  p.kt_per_split = (nkt + ks - 1) / ks;
  p.ksplit = (nkt + p.kt_per_split - 1) / p.kt_per_split;
  // Grids a little above a multiple of the 512 resident blocks (PatchGAN: 580 and 529 tiles): splitting EVERY tile
  // along K pays slab traffic for all of them.  Instead the whole rounds run unsplit and only the m-tiles of the last,
  // partly filled round are split, finely enough to fill that round.
  static const int tail_on = getenv("CSG_IGEMM_TAIL_SPLIT") ? atoi(getenv("CSG_IGEMM_TAIL_SPLIT")) : 1;
  if (tail_on && force == 0 && blocks > 512 && blocks < 4096 && nkt >= 16 && p.d.Cout % 4 == 0) {
    const int rounds = blocks / 512;                                  // whole rounds
    int full_mt = (rounds * 512) / p.ntiles;                          // m-tiles that fit in them
    const int tail_tiles = (p.mtiles - full_mt) * p.ntiles;
    if (full_mt > 0 && tail_tiles > 0 && tail_tiles <= 320) {
      int tks = 512 / tail_tiles;
      if (tks > nkt / 8) tks = nkt / 8;
      if (tks >= 2) {
        p.ksplit = 1;
        p.kt_per_split = nkt;
        p.tail_tile0 = full_mt * p.ntiles;
        p.tail_kt_per = (nkt + tks - 1) / tks;
        p.tail_ks = (nkt + p.tail_kt_per - 1) / p.tail_kt_per;
        p.tail_m0 = full_mt * IG_BM;
        if (p.tail_ks < 2) p.tail_ks = p.tail_tile0 = p.tail_kt_per = p.tail_m0 = 0;
      }
    }
  }
}

static int64_t fwd_slab_floats(const IgemmParams& p) {
  if (p.tail_ks > 1) return (int64_t)p.tail_ks * (p.M - p.tail_m0) * p.d.Cout;
  return p.ksplit > 1 ? (int64_t)p.ksplit * p.M * p.d.Cout : 0;
}
static int64_t fwd_grid(const IgemmParams& p) {
  if (p.tail_ks > 1) return (int64_t)p.tail_tile0 + (int64_t)(p.mtiles * p.ntiles - p.tail_tile0) * p.tail_ks;
  return (int64_t)p.mtiles * p.ntiles * p.ksplit;
}

template <int BN>
static int launch_fwd(IgemmParams& p, const float* x, const float* w, const float* bias, const float* res, float* y,
                      float* ws, hipStream_t s) {
  static bool attr_set = false;
  size_t shm = (size_t)(2 * IG_BM * IG_LD + 2 * BN * IG_LD) * 4 + 48 * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_igemm_fwd<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_conv_fwd: hipFuncSetAttribute(%zu bytes of LDS): %s", shm, hipGetErrorString(e));
    attr_set = true;
  }
  CSG_LAUNCH(k_igemm_fwd<BN>, dim3((unsigned)fwd_grid(p)), dim3(256), shm, s, p, x, w, bias, res, y, ws);
  return check_launch("csg_conv_fwd");
}

static void wgrad_plan(const IgemmParams& p, int bi, int& itiles, int& jtiles, int& nsplit, int& cps) {
  itiles = (p.d.Cout + bi - 1) / bi;
  jtiles = (p.Ktot + 127) / 128;
  const int nch = (p.M + 31) / 32;
  nsplit = pick_split(itiles * jtiles, nch, 8, 256);
  cps = (nch + nsplit - 1) / nsplit;
  nsplit = (nch + cps - 1) / cps;  // no empty splits
}

template <int BI>
static int launch_wgrad(IgemmParams& p, const float* x, const float* dy, float* out, int itiles, int jtiles,
                        int nsplit, int cps, float* dbp, hipStream_t s) {
  static bool attr_set = false;
  size_t shm = (size_t)(2 * 32 * BI + 2 * 32 * WG_LDB) * 4 + 48 * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_igemm_wgrad<BI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_conv_bwd_weight: hipFuncSetAttribute(%zu bytes of LDS): %s", shm, hipGetErrorString(e));
    attr_set = true;
  }
  CSG_LAUNCH(k_igemm_wgrad<BI>, dim3((unsigned)(itiles * jtiles * nsplit)), dim3(256), shm, s, p, x, dy, out,
                     itiles, jtiles, nsplit, cps, dbp);
  return check_launch("csg_conv_bwd_weight");
}

static int pick_bi(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

// ---- several descriptors (same weights, input, output tensor and Cout), one launch -------------------------------
static int multi_plan(const csg_conv_desc* descs, int n, IgemmMulti& mp, int64_t& ws_floats, int& max_blocks,
                      const char* who) {
  CSG_REQUIRE(descs != nullptr && n >= 1 && n <= IG_MAXCLS, CSG_E_BADSHAPE, "%s: 1..%d descriptors", who, IG_MAXCLS);
  int64_t blocks_total = 0;
  for (int c = 0; c < n; ++c) {
    int rc = validate(&descs[c], who);
    if (rc) return rc;
    CSG_REQUIRE(descs[c].Cout == descs[0].Cout && descs[c].Cin == descs[0].Cin && descs[c].x_cs == descs[0].x_cs &&
                    descs[c].y_cs == descs[0].y_cs && descs[c].wtaps == descs[0].wtaps && descs[c].B == descs[0].B,
                CSG_E_BADSHAPE, "%s: the descriptors must share channels, strides, weights and batch", who);
    fill(mp.p[c], &descs[c]);
    const int bn = pick_bn(descs[c].Cout);
    mp.p[c].mtiles = (mp.p[c].M + IG_BM - 1) / IG_BM;
    mp.p[c].ntiles = (descs[c].Cout + bn - 1) / bn;
    blocks_total += (int64_t)mp.p[c].mtiles * mp.p[c].ntiles;
  }
  for (int c = n; c < IG_MAXCLS; ++c) mp.p[c] = mp.p[0];
  // one split factor for all classes, chosen for the grid they form TOGETHER
  ws_floats = 0;
  max_blocks = 0;
  for (int c = 0; c < n; ++c) {
    IgemmParams& p = mp.p[c];
    const int nkt = (p.Ktot + IG_BK - 1) / IG_BK;
    int ks = 1;
    if (blocks_total < 1024 && nkt >= 16 && p.d.Cout % 4 == 0) ks = pick_split((int)blocks_total, nkt, 8, 64);
    p.kt_per_split = (nkt + ks - 1) / ks;
    p.ksplit = (nkt + p.kt_per_split - 1) / p.kt_per_split;
    p.ws_off = ws_floats;
    if (p.ksplit > 1) ws_floats += (int64_t)p.ksplit * p.M * p.d.Cout;
    const int blocks = p.mtiles * p.ntiles * p.ksplit;
    if (blocks > max_blocks) max_blocks = blocks;
  }
  return CSG_OK;
}

template <int BN>
static int launch_fwd_multi(IgemmMulti& mp, int n, int max_blocks, const float* x, const float* w, const float* bias,
                            const float* res, float* y, float* ws, hipStream_t s) {
  static bool attr_set = false;
  size_t shm = (size_t)(2 * IG_BM * IG_LD + 2 * BN * IG_LD) * 4 + 48 * 4;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_igemm_fwd<BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    CSG_REQUIRE(e == hipSuccess, CSG_E_LAUNCH, "csg_conv_fwd_multi: hipFuncSetAttribute(%zu bytes of LDS): %s", shm, hipGetErrorString(e));
    attr_set = true;
  }
  CSG_LAUNCH((k_igemm_fwd<BN, true>), dim3((unsigned)max_blocks, (unsigned)n), dim3(256), shm, s, mp, x, w, bias, res,
                     y, ws);
  return check_launch("csg_conv_fwd_multi");
}

extern "C" {

int64_t csg_conv_fwd_workspace(const csg_conv_desc* d) {
  if (validate(d, "csg_conv_fwd_workspace")) return -1;
  IgemmParams p;
  fill(p, d);
  fwd_plan(p);
  return fwd_slab_floats(p) * 4;
}

int csg_conv_fwd(const csg_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                 float* y, float* workspace, int64_t workspace_bytes, void* stream) {
  int rc = validate(d, "csg_conv_fwd");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_fwd: x and w must be 16-byte aligned");
  CSG_REQUIRE(!d->res_gate || (d->act == CSG_ACT_NONE && residual != nullptr && !d->accumulate), CSG_E_UNSUPPORTED,
              "csg_conv_fwd: res_gate needs act == NONE, a residual (the gating tensor) and no accumulation");
  IgemmParams p;
  fill(p, d);
  fwd_plan(p);
  const int64_t need = fwd_slab_floats(p) * 4;
  CSG_REQUIRE(need == 0 || (workspace != nullptr && workspace_bytes >= need), CSG_E_WORKSPACE,
              "csg_conv_fwd: needs %lld bytes of workspace (csg_conv_fwd_workspace), got %lld", (long long)need,
              (long long)workspace_bytes);
  hipStream_t s = (hipStream_t)stream;
  const int bn = pick_bn(d->Cout);
  {
    // algorithmic FLOPs: 2 * M * Ktot * Cout (zero-padding taps included, as FlopCounterMode counts them)
    ProfScope ps(bn == 128 ? K_IGEMM_FWD : K_IGEMM_FWD64, 2.0 * p.M * (double)p.Ktot * d->Cout, s);
    if (bn == 32)
      rc = launch_fwd<32>(p, x, w, bias, residual, y, workspace, s);
    else if (bn == 64)
      rc = launch_fwd<64>(p, x, w, bias, residual, y, workspace, s);
    else
      rc = launch_fwd<128>(p, x, w, bias, residual, y, workspace, s);
    if (rc) return rc;
  }
  if (p.ksplit > 1 || p.tail_ks > 1) {
    ProfScope ps(K_SPLITK_EPI, (double)fwd_slab_floats(p) * 4 + (double)(p.M - p.tail_m0) * d->Cout * 4, s);
    const int64_t n4 = (int64_t)(p.M - p.tail_m0) * d->Cout / 4;
    int64_t g = cdiv(n4, 256);
    if (g > 4096) g = 4096;
    CSG_LAUNCH(k_splitk_epilogue<false>, dim3((unsigned)g), dim3(256), 0, s, p, workspace, bias, residual, y);
    rc = check_launch("csg_conv_fwd(split-K epilogue)");
  }
  return rc;
}

int64_t csg_conv_fwd_multi_workspace(const csg_conv_desc* descs, int32_t n) {
  IgemmMulti mp;
  int64_t wsf = 0;
  int mb = 0;
  if (multi_plan(descs, n, mp, wsf, mb, "csg_conv_fwd_multi_workspace")) return -1;
  return wsf * 4;
}

int csg_conv_fwd_multi(const csg_conv_desc* descs, int32_t n, const float* x, const float* w, const float* bias,
                       const float* residual, float* y, float* workspace, int64_t workspace_bytes, void* stream) {
  IgemmMulti mp;
  int64_t wsf = 0;
  int mb = 0;
  int rc = multi_plan(descs, n, mp, wsf, mb, "csg_conv_fwd_multi");
  if (rc) return rc;
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_fwd_multi: x and w must be 16-byte aligned");
  CSG_REQUIRE(wsf == 0 || (workspace != nullptr && workspace_bytes >= wsf * 4), CSG_E_WORKSPACE,
              "csg_conv_fwd_multi: needs %lld bytes of workspace", (long long)(wsf * 4));
  hipStream_t s = (hipStream_t)stream;
  const int bn = pick_bn(descs[0].Cout);
  double flops = 0.0;
  bool any_split = false;
  int64_t max_n4 = 0;
  for (int c = 0; c < n; ++c) {
    flops += 2.0 * mp.p[c].M * (double)mp.p[c].Ktot * descs[c].Cout;
    any_split = any_split || mp.p[c].ksplit > 1;
    const int64_t n4 = (int64_t)mp.p[c].M * descs[c].Cout / 4;
    if (n4 > max_n4) max_n4 = n4;
  }
  {
    ProfScope ps(bn == 128 ? K_IGEMM_FWD : K_IGEMM_FWD64, flops, s);
    if (bn == 32)
      rc = launch_fwd_multi<32>(mp, n, mb, x, w, bias, residual, y, workspace, s);
    else if (bn == 64)
      rc = launch_fwd_multi<64>(mp, n, mb, x, w, bias, residual, y, workspace, s);
    else
      rc = launch_fwd_multi<128>(mp, n, mb, x, w, bias, residual, y, workspace, s);
    if (rc) return rc;
  }
  if (any_split) {
    ProfScope ps(K_SPLITK_EPI, (double)wsf * 4, s);
    int64_t g = cdiv(max_n4, 256);
    if (g > 2048) g = 2048;
    CSG_LAUNCH(k_splitk_epilogue<true>, dim3((unsigned)g, (unsigned)n), dim3(256), 0, s, mp, workspace, bias, residual,
                       y);
    rc = check_launch("csg_conv_fwd_multi(split-K epilogue)");
  }
  return rc;
}

int64_t csg_conv_bwd_weight_workspace(const csg_conv_desc* d) {
  if (validate(d, "csg_conv_bwd_weight_workspace")) return -1;
  IgemmParams p;
  fill(p, d);
  int it, jt, ns, cps;
  wgrad_plan(p, pick_bi(d->Cout), it, jt, ns, cps);
  return ns > 1 ? (int64_t)ns * d->Cout * (p.wrow + 1) * 4 : 0;      // dW slabs + one bias-gradient row per split
}

int csg_conv_bwd_weight(const csg_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                        float* workspace, int64_t workspace_bytes, void* stream) {
  int rc = validate(d, "csg_conv_bwd_weight");
  if (rc) return rc;
  CSG_REQUIRE(d->Cout % 4 == 0 && d->y_cs % 4 == 0, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: Cout=%d and y_cs=%d must be multiples of 4", d->Cout, d->y_cs);
  CSG_REQUIRE(d->os == 1 && d->ooy == 0 && d->oox == 0 && d->OHg == d->OHf && d->OWg == d->OWf, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: needs the forward descriptor (dense output grid)");
  CSG_REQUIRE(d->ntaps == d->wtaps, CSG_E_UNSUPPORTED, "csg_conv_bwd_weight: every weight tap must be listed");
  CSG_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0 && ((uintptr_t)dw % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: pointers must be 16-byte aligned");
  CSG_REQUIRE((int64_t)32 * d->y_cs * 4 < MAX_RECORDS, CSG_E_UNSUPPORTED, "csg_conv_bwd_weight: y_cs too large");
  IgemmParams p;
  fill(p, d);
  const int bi = pick_bi(d->Cout);
  int itiles, jtiles, nsplit, cps;
  wgrad_plan(p, bi, itiles, jtiles, nsplit, cps);
  // a split's pixel range must stay within 32-bit offsets of its first image
  {
    const int64_t npix = (int64_t)d->OHg * d->OWg;
    const int64_t span = (((int64_t)cps * 32 + npix - 1) / npix + 1) * (int64_t)p.img_bytes;
    CSG_REQUIRE(span < MAX_RECORDS, CSG_E_UNSUPPORTED, "csg_conv_bwd_weight: split spans more than 2 GB of input");
  }
  const int64_t need = nsplit > 1 ? (int64_t)nsplit * d->Cout * (p.wrow + 1) * 4 : 0;
  CSG_REQUIRE(db == nullptr || ((uintptr_t)db % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_conv_bwd_weight: db must be 16-byte aligned");
  CSG_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr), CSG_E_WORKSPACE,
              "csg_conv_bwd_weight: workspace %ld < %ld bytes", (long)workspace_bytes, (long)need);
  hipStream_t s = (hipStream_t)stream;
  {
    ProfScope ps(K_IGEMM_WGRAD, 2.0 * p.M * (double)p.Ktot * d->Cout, s);
    float* out = nsplit > 1 ? workspace : dw;
    float* dbp = db == nullptr ? nullptr : (nsplit > 1 ? workspace + (int64_t)nsplit * d->Cout * p.wrow : db);
    if (bi == 32)
      rc = launch_wgrad<32>(p, x, dy, out, itiles, jtiles, nsplit, cps, dbp, s);
    else if (bi == 64)
      rc = launch_wgrad<64>(p, x, dy, out, itiles, jtiles, nsplit, cps, dbp, s);
    else
      rc = launch_wgrad<128>(p, x, dy, out, itiles, jtiles, nsplit, cps, dbp, s);
    if (rc) return rc;
  }
  if (nsplit > 1) {
    const int64_t n4 = (int64_t)d->Cout * p.wrow / 4;
    ProfScope ps(K_WGRAD_REDUCE, (double)(nsplit + 1) * n4 * 16, s);
    launch_slab_reduce(workspace, n4 * 4, dw, workspace + (int64_t)nsplit * d->Cout * p.wrow, db != nullptr ? d->Cout : 0,
                       db, nsplit, s);
    rc = check_launch("csg_conv_bwd_weight(reduce)");
  }
  return rc;
}

}  // extern "C"
