// K6 — boxes_to_layout as ONE pass over the output (reference: sg2im/layout.py:12-45,80-112,
// 156-188; called per sample from spade/models/networks/generator.py:82-96 and
// discriminator.py:102-119).
//
// The reference materialises grid_sample(constant (O,D,8,8) image) = an (O,D,H,W) tensor and
// scatter_adds it: O times the output in HBM traffic.  grid_sample of a constant image with zero
// padding factorises into a separable coverage weight, so
//     layout[b,y,x,d] = sum_o vec[b,o,d] * cy[o,y] * cx[o,x]        (objects in index order)
// and the kernel writes each output byte exactly once (HBM-bound: S*H*W*4 bytes per image).
// Output is NHWC; it may be a channel slice of a wider buffer (the discriminator's input), and it
// may be a nearest-neighbour down-sampled view (the SPADE seg pyramid).
#include "csg_common.h"

using namespace csg;

// torch.linspace(0,1,n)[i] exactly as ATen evaluates it (symmetric halves)
__device__ __forceinline__ float lin01(int i, int n) {
  if (n <= 1) return 0.f;
  float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? (float)i * step : 1.0f - (float)(n - 1 - i) * step;
}

// coverage of one box along one axis at pixel-centre coordinate t (see oracle/functional.py
// box_coverage): _boxes_to_grid (layout.py:98-110) then bilinear grid_sample, align_corners=False,
// zeros padding, on an 8-pixel constant line.
__device__ __forceinline__ float coverage(float t, float lo, float size) {
  float g = ((t - lo) / size) * 2.0f - 1.0f;
  float ix = ((g + 1.0f) * 8.0f - 1.0f) / 2.0f;
  float i0 = floorf(ix);
  float fr = ix - i0;
  float w0 = (i0 >= 0.0f && i0 <= 7.0f) ? (1.0f - fr) : 0.0f;
  float w1 = (i0 + 1.0f >= 0.0f && i0 + 1.0f <= 7.0f) ? fr : 0.0f;
  return w0 + w1;
}

// support test for an n-pixel source line (coverage() is the n = 8 case of boxes_to_layout)
__device__ __forceinline__ float coverage_n(float t, float lo, float size, int n) {
  float g = ((t - lo) / size) * 2.0f - 1.0f;
  float ix = ((g + 1.0f) * (float)n - 1.0f) / 2.0f;
  float i0 = floorf(ix);
  float fr = ix - i0;
  float w0 = (i0 >= 0.0f && i0 <= (float)(n - 1)) ? (1.0f - fr) : 0.0f;
  float w1 = (i0 + 1.0f >= 0.0f && i0 + 1.0f <= (float)(n - 1)) ? fr : 0.0f;
  return w0 + w1;
}

// Bilinear tap of grid_sample(align_corners=False, zeros padding) along one axis of an n-pixel
// image: returns the lower tap index i0 (may be -1 .. n-1), its weight w0 and the upper tap's w1,
// each already zeroed when the tap is outside the image.
__device__ __forceinline__ void axis_taps(float t, float lo, float size, int n, int& i0, float& w0, float& w1) {
  float g = ((t - lo) / size) * 2.0f - 1.0f;
  float ix = ((g + 1.0f) * (float)n - 1.0f) / 2.0f;
  float f0 = floorf(ix);
  float fr = ix - f0;
  // clamp before the int conversion: far-away boxes give huge |ix|
  f0 = fminf(fmaxf(f0, -2.0f), (float)n);
  i0 = (int)f0;
  w0 = (i0 >= 0 && i0 <= n - 1) ? (1.0f - fr) : 0.0f;
  w1 = (i0 + 1 >= 0 && i0 + 1 <= n - 1) ? fr : 0.0f;
}

// weight of one object at one output pixel when a (M,M) mask modulates it (masks_to_layout,
// sg2im/layout.py:48-77): bilinear sample of the mask; the row taps (iy0, wy0, wy1) are precomputed
__device__ __forceinline__ float mask_weight(const float* __restrict__ mk, int M, int iy0, float wy0, float wy1,
                                             float tx, float x0, float ww) {
  int ix0;
  float wx0, wx1;
  axis_taps(tx, x0, ww, M, ix0, wx0, wx1);
  float acc = 0.f;
  if (wy0 != 0.f) {
    const float* row = mk + iy0 * M;
    if (wx0 != 0.f) acc += row[ix0] * wy0 * wx0;
    if (wx1 != 0.f) acc += row[ix0 + 1] * wy0 * wx1;
  }
  if (wy1 != 0.f) {
    const float* row = mk + (iy0 + 1) * M;
    if (wx0 != 0.f) acc += row[ix0] * wy1 * wx0;
    if (wx1 != 0.f) acc += row[ix0 + 1] * wy1 * wx1;
  }
  return acc;
}

// d coverage / d ix of grid_sample's bilinear interpolation on a constant n-pixel line (what ATen's
// grid_sampler backward accumulates into the grid gradient): +1 while only the upper tap is inside the line
// (ix in [-1, 0)), -1 while only the lower one is (ix in [n-1, n)), 0 elsewhere.
__device__ __forceinline__ float coverage_dix(float t, float lo, float size, int n) {
  float g = ((t - lo) / size) * 2.0f - 1.0f;
  float ix = ((g + 1.0f) * (float)n - 1.0f) / 2.0f;
  float i0 = floorf(ix);
  float v0 = (i0 >= 0.0f && i0 <= (float)(n - 1)) ? 1.0f : 0.0f;
  float v1 = (i0 + 1.0f >= 0.0f && i0 + 1.0f <= (float)(n - 1)) ? 1.0f : 0.0f;
  return v1 - v0;
}

// mask_weight together with its derivatives w.r.t. the un-normalised sample coordinates (ix, iy)
__device__ __forceinline__ void mask_weight_grad(const float* __restrict__ mk, int M, float ty, float y0, float hh,
                                                 float tx, float x0, float ww, float& w, float& dwdix, float& dwdiy) {
  float gy = ((ty - y0) / hh) * 2.0f - 1.0f, gx = ((tx - x0) / ww) * 2.0f - 1.0f;
  float fy = ((gy + 1.0f) * (float)M - 1.0f) / 2.0f, fx = ((gx + 1.0f) * (float)M - 1.0f) / 2.0f;
  float y0f = floorf(fy), x0f = floorf(fx);
  float ry = fy - y0f, rx = fx - x0f;
  y0f = fminf(fmaxf(y0f, -2.0f), (float)M);
  x0f = fminf(fmaxf(x0f, -2.0f), (float)M);
  const int iy = (int)y0f, ix = (int)x0f;
  float m[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int yy = iy + a, xx = ix + c;
      m[a][c] = (yy >= 0 && yy < M && xx >= 0 && xx < M) ? mk[yy * M + xx] : 0.f;
    }
  w = m[0][0] * (1.f - ry) * (1.f - rx) + m[0][1] * (1.f - ry) * rx + m[1][0] * ry * (1.f - rx) + m[1][1] * ry * rx;
  dwdix = (m[0][1] - m[0][0]) * (1.f - ry) + (m[1][1] - m[1][0]) * ry;
  dwdiy = (m[1][0] - m[0][0]) * (1.f - rx) + (m[1][1] - m[0][1]) * rx;
}

#define LAY_OB 32    // active objects staged in LDS at a time
#define LAY_CULL 256 // objects culled and compacted per pass (one per thread)
#define LAY_PXC 256  // max pixels per block chunk
#define LAY_EPT 8    // float4 elements per thread

// The image discriminator's input cat([img, layout]) (discriminator.py:120) in one pass: with `img` set, the thread that
// owns channel quad 0 of a pixel also writes the channels behind the layout — [img(3) | zeros] up to the pixel stride.
struct LayTail {
  const float* img;        // (B,3,H,W) with element strides sb, sc, sh, sw; nullptr: no tail
  long long sb, sc, sh, sw;
  int nquads;              // float4 quads per pixel behind the S layout channels (out_cs - S) / 4
};
__device__ __forceinline__ void lay_write_tail(const LayTail& t, float* px, int S, int b, int y, int x) {
  const float* ip = t.img + b * t.sb + y * t.sh + x * t.sw;
  *(float4*)(px + S) = make_float4(ip[0], ip[t.sc], ip[2 * t.sc], 0.f);
  for (int q = 1; q < t.nquads; ++q) *(float4*)(px + S + 4 * q) = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ __launch_bounds__(256) void k_layout_fwd(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                     const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ masks, int M, int O, int S, int H,
                                                     int W, int OH, int OW, int pxc, float* __restrict__ out,
                                                     int out_cs, int out_off, LayTail tail) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wx = sm;                       // [LAY_OB][pxc]
  float* s_vec = sm + LAY_OB * pxc;       // [LAY_OB][S]
  float* s_wy = s_vec + LAY_OB * S;       // [LAY_CULL] row weight of each active slot
  float* s_wy1 = s_wy + LAY_CULL;         // [LAY_CULL] mask path: weight of the upper row tap
  int* s_iy0 = (int*)(s_wy1 + LAY_CULL);  // [LAY_CULL] mask path: lower row tap index
  int* s_act = s_iy0 + LAY_CULL;          // [LAY_CULL] object index of each active slot
  int* s_cnt = s_act + LAY_CULL;          // [4] active objects per wave

  const int tid = threadIdx.x;
  const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * pxc;
  const int npx = min(pxc, OW - x0);
  const int qpp = S >> 2;
  const int nel = npx * qpp;
  const int ysrc = min((int)(((int64_t)y * H) / OH), H - 1);
  const float ty = lin01(ysrc, H);

  float4 acc[LAY_EPT];
  int eq[LAY_EPT], ex[LAY_EPT];
  // S/4 dividing 256 (S = 32, 128: every configuration of the trainer): a thread owns ONE channel quad of EIGHT
  // CONSECUTIVE pixels, so an object costs it one vector read + two weight-quad reads for 32 FMAs instead of 16 reads
  const bool blocked = (256 % qpp) == 0 && (pxc % LAY_EPT) == 0;
#pragma unroll
  for (int i = 0; i < LAY_EPT; ++i) {
    acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = tid + 256 * i;
    ex[i] = e / qpp;
    eq[i] = e - ex[i] * qpp;
    if (blocked) {
      eq[i] = tid % qpp;
      ex[i] = (tid / qpp) * LAY_EPT + i;
    }
  }
  const float* bx = boxes + (int64_t)b * O * 4;
  const uint8_t* vb = valid + (int64_t)b * O;
  const float* vv = vecs + (int64_t)b * O * S;

  // Culling: ONE pass over up to 256 objects (a thread each, all four waves), compacted in index order — the order
  // of the reference's sum.  Dense scenes then stage their ~10 surviving objects in one batch instead of walking four
  // 32-object batches with three barriers and two dependent global-load latencies each.
  for (int ob = 0; ob < O; ob += LAY_CULL) {
    __syncthreads();
    {
      const int o = ob + tid;
      float wy = 0.f, wy1 = 0.f;
      bool act = false;
      int iy0 = 0;
      if (o < O && vb[o]) {
        const float4 bq = *(const float4*)(bx + o * 4);
        if (masks == nullptr) {
          wy = coverage(ty, bq.y, bq.w);
          act = (wy != 0.0f);
        } else {
          axis_taps(ty, bq.y, bq.w, M, iy0, wy, wy1);
          act = (wy != 0.0f) || (wy1 != 0.0f);
        }
        // also drop the objects whose x support misses this block's pixel chunk.  The bilinear weight vanishes outside
        // x0 - w/(2n) < t < x0 + w (1 + 1/(2n))  (n source pixels); the test keeps one output pixel of slack on each
        // side, and a skipped object would have added exact zeros, so the sum — and its order among the remaining
        // objects — is unchanged.
        if (act) {
          const float bx0 = bq.x, bw = bq.z;
          const float n = masks == nullptr ? 8.0f : (float)M;
          const float lo = fminf(bx0 - bw / (2.0f * n), bx0 + bw * (1.0f + 1.0f / (2.0f * n)));
          const float hi = fmaxf(bx0 - bw / (2.0f * n), bx0 + bw * (1.0f + 1.0f / (2.0f * n)));
          const int xs0 = min((int)(((int64_t)x0 * W) / OW), W - 1);
          const int xs1 = min((int)(((int64_t)(x0 + npx - 1) * W) / OW), W - 1);
          const float step = W > 1 ? 1.0f / (float)(W - 1) : 1.0f;
          // (NaN boxes fail neither comparison: they stay active, as in the reference)
          if (lin01(xs1, W) + step < lo || lin01(xs0, W) - step > hi) act = false;
        }
      }
      const unsigned long long m = __ballot(act);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) s_cnt[wv] = __popcll(m);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wv; ++w) base += s_cnt[w];
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (act) {
        s_wy[slot] = wy;
        s_wy1[slot] = wy1;
        s_iy0[slot] = iy0;
        s_act[slot] = o;
      }
    }
    __syncthreads();
    const int ntot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    for (int a0 = 0; a0 < ntot; a0 += LAY_OB) {
    const int nact = min(LAY_OB, ntot - a0);
    if (a0 > 0) __syncthreads();                 // the previous group's s_wx / s_vec are still being read
      for (int i = tid; i < nact * npx; i += 256) {
        const int a = i / npx, xl = i - a * npx;
        const int o = s_act[a0 + a];
        const int xsrc = min((int)(((int64_t)(x0 + xl) * W) / OW), W - 1);
        if (masks == nullptr)
          s_wx[a * pxc + xl] = coverage(lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]);
        else  // full 2-D weight of this pixel; the row factor below is 1
          s_wx[a * pxc + xl] = mask_weight(masks + ((int64_t)b * O + o) * M * M, M, s_iy0[a0 + a], s_wy[a0 + a],
                                           s_wy1[a0 + a], lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]);
      }
      for (int i = tid; i < nact * (S >> 2); i += 256) {
        const int a = i / (S >> 2), d4 = i - a * (S >> 2);
        *(float4*)&s_vec[a * S + d4 * 4] = *(const float4*)&vv[(int64_t)s_act[a0 + a] * S + d4 * 4];
      }
      __syncthreads();
      if (blocked) {
        const int px0 = (tid / qpp) * LAY_EPT, q4 = (tid % qpp) * 4;
        if (px0 < npx) {
          for (int a = 0; a < nact; ++a) {
            const float wy = masks == nullptr ? s_wy[a0 + a] : 1.0f;
            const float4 v = *(const float4*)&s_vec[a * S + q4];
            const float4 w0 = *(const float4*)&s_wx[a * pxc + px0], w1 = *(const float4*)&s_wx[a * pxc + px0 + 4];
            const float wv[LAY_EPT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int i = 0; i < LAY_EPT; ++i) {
              const float w = wy * wv[i];
              acc[i].x += v.x * w;
              acc[i].y += v.y * w;
              acc[i].z += v.z * w;
              acc[i].w += v.w * w;
            }
          }
        }
        continue;
      }
      for (int a = 0; a < nact; ++a) {
        const float wy = masks == nullptr ? s_wy[a0 + a] : 1.0f;
#pragma unroll
        for (int i = 0; i < LAY_EPT; ++i) {
          if (tid + 256 * i < nel) {
            float w = wy * s_wx[a * pxc + ex[i]];
            float4 v = *(const float4*)&s_vec[a * S + eq[i] * 4];
            acc[i].x += v.x * w;
            acc[i].y += v.y * w;
            acc[i].z += v.z * w;
            acc[i].w += v.w * w;
          }
        }
      }
    }
  }
  float* orow = out + ((int64_t)(b * OH + y) * OW + x0) * out_cs + out_off;
#pragma unroll
  for (int i = 0; i < LAY_EPT; ++i) {
    if (blocked ? ex[i] < npx : tid + 256 * i < nel) {
      *(float4*)&orow[(int64_t)ex[i] * out_cs + eq[i] * 4] = acc[i];
      if (tail.img != nullptr && eq[i] == 0) lay_write_tail(tail, orow + (int64_t)ex[i] * out_cs, S, b, y, x0 + ex[i]);
    }
  }
}

// The same sum for boxes_to_layout (no masks) with S/4 dividing 256, ROWS output rows per block: the x coverage and
// the object vectors do not depend on the row, so they are staged once for ROWS rows (dense scenes spend their time
// culling and staging, not accumulating).  A thread owns one channel quad of eight consecutive pixels in each row.  An
// object that covers only some of the ROWS rows adds exact zeros to the others: the per-pixel sum and its order are
// those of k_layout_fwd.
template <int ROWS>
__global__ __launch_bounds__(256) void k_layout_fwd_rows(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                          const uint8_t* __restrict__ valid, int O, int S, int H, int W,
                                                          int OH, int OW, int pxc, float* __restrict__ out, int out_cs,
                                                          int out_off, LayTail tail) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wx = sm;                         // [LAY_OB][pxc]
  float* s_vec = sm + LAY_OB * pxc;         // [LAY_OB][S]
  float* s_wy = s_vec + LAY_OB * S;         // [LAY_CULL][ROWS]
  int* s_act = (int*)(s_wy + LAY_CULL * ROWS);  // [LAY_CULL]
  int* s_cnt = s_act + LAY_CULL;            // [4]
  const int tid = threadIdx.x;
  const int b = blockIdx.z, y0 = blockIdx.y * ROWS, x0 = blockIdx.x * pxc;
  const int npx = min(pxc, OW - x0);
  const int qpp = S >> 2;
  const int px0 = (tid / qpp) * LAY_EPT, q4 = (tid % qpp) * 4;
  float ty[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) ty[r] = lin01(min((int)(((int64_t)min(y0 + r, OH - 1) * H) / OH), H - 1), H);
  float4 acc[ROWS][LAY_EPT];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < LAY_EPT; ++i) acc[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* bx = boxes + (int64_t)b * O * 4;
  const uint8_t* vb = valid + (int64_t)b * O;
  const float* vv = vecs + (int64_t)b * O * S;
  const int xs0 = min((int)(((int64_t)x0 * W) / OW), W - 1);
  const int xs1 = min((int)(((int64_t)(x0 + npx - 1) * W) / OW), W - 1);
  const float step = W > 1 ? 1.0f / (float)(W - 1) : 1.0f;
  const float tx_lo = lin01(xs0, W) - step, tx_hi = lin01(xs1, W) + step;

  for (int ob = 0; ob < O; ob += LAY_CULL) {
    __syncthreads();
    {
      const int o = ob + tid;
      float wy[ROWS];
      bool act = false;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) wy[r] = 0.f;
      if (o < O && vb[o]) {
        const float4 bq = *(const float4*)(bx + o * 4);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          wy[r] = (y0 + r < OH) ? coverage(ty[r], bq.y, bq.w) : 0.f;
          act = act || (wy[r] != 0.0f);
        }
        if (act) {                          // x support vs this block's pixel chunk (see k_layout_fwd)
          const float lo = fminf(bq.x - bq.z / 16.0f, bq.x + bq.z * (1.0f + 1.0f / 16.0f));
          const float hi = fmaxf(bq.x - bq.z / 16.0f, bq.x + bq.z * (1.0f + 1.0f / 16.0f));
          if (tx_hi < lo || tx_lo > hi) act = false;
        }
      }
      const unsigned long long m = __ballot(act);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) s_cnt[wv] = __popcll(m);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wv; ++w) base += s_cnt[w];
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (act) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s_wy[slot * ROWS + r] = wy[r];
        s_act[slot] = o;
      }
    }
    __syncthreads();
    const int ntot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    for (int a0 = 0; a0 < ntot; a0 += LAY_OB) {
      const int nact = min(LAY_OB, ntot - a0);
      if (a0 > 0) __syncthreads();
      for (int i = tid; i < nact * npx; i += 256) {
        const int a = i / npx, xl = i - a * npx;
        const int o = s_act[a0 + a];
        const int xsrc = min((int)(((int64_t)(x0 + xl) * W) / OW), W - 1);
        s_wx[a * pxc + xl] = coverage(lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]);
      }
      for (int i = tid; i < nact * qpp; i += 256) {
        const int a = i / qpp, d4 = i - a * qpp;
        *(float4*)&s_vec[a * S + d4 * 4] = *(const float4*)&vv[(int64_t)s_act[a0 + a] * S + d4 * 4];
      }
      __syncthreads();
      if (px0 < npx) {
        for (int a = 0; a < nact; ++a) {
          const float4 v = *(const float4*)&s_vec[a * S + q4];
          const float4 w0 = *(const float4*)&s_wx[a * pxc + px0], w1 = *(const float4*)&s_wx[a * pxc + px0 + 4];
          const float wv[LAY_EPT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float wy = s_wy[(a0 + a) * ROWS + r];
#pragma unroll
            for (int i = 0; i < LAY_EPT; ++i) {
              const float w = wy * wv[i];
              acc[r][i].x += v.x * w;
              acc[r][i].y += v.y * w;
              acc[r][i].z += v.z * w;
              acc[r][i].w += v.w * w;
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    if (y0 + r >= OH) break;
    float* orow = out + ((int64_t)(b * OH + y0 + r) * OW + x0) * out_cs + out_off;
#pragma unroll
    for (int i = 0; i < LAY_EPT; ++i)
      if (px0 + i < npx) {
        *(float4*)&orow[(int64_t)(px0 + i) * out_cs + q4] = acc[r][i];
        if (tail.img != nullptr && q4 == 0) lay_write_tail(tail, orow + (int64_t)(px0 + i) * out_cs, S, b, y0 + r, x0 + px0 + i);
      }
  }
}

// ---- backward for boxes_to_layout without box gradients, dense scenes --------------------------------------------
// k_layout_bwd gives every (object, image) a block that walks the object's own box support: the gradient map is read
// once per covering object (~7x on config C5).  Here it is read ONCE: pass 1 has the forward's tiling (ROWS rows x one
// pixel chunk per block, the tile of dout held in registers) and, for each object active in the tile, reduces
// sum_pixels weight * dout over the tile — partial[b][tile][o][S]; it also records which (tile, object) pairs it wrote.
// Pass 2 (one block per (object, image)) adds an object's partials in tile order.  Fixed association everywhere:
// bit-reproducible, no atomics.
#define LAY_BB 8      // objects reduced per LDS round of pass 1
template <int ROWS>
__global__ __launch_bounds__(256) void k_layout_bwd_tiles(const float* __restrict__ dout, int out_cs, int out_off,
                                                           const float* __restrict__ boxes,
                                                           const uint8_t* __restrict__ valid, int O, int S, int H, int W,
                                                           int OH, int OW, int pxc, float* __restrict__ partial,
                                                           uint8_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float4* s_part = (float4*)sm;                       // [LAY_BB][256]
  float* s_wx = sm + LAY_BB * 256 * 4;                // [LAY_BB][pxc]
  float* s_wy = s_wx + LAY_BB * pxc;                  // [LAY_CULL][ROWS]
  int* s_act = (int*)(s_wy + LAY_CULL * ROWS);        // [LAY_CULL]
  int* s_cnt = s_act + LAY_CULL;                      // [4]
  const int tid = threadIdx.x;
  const int b = blockIdx.z, y0 = blockIdx.y * ROWS, x0 = blockIdx.x * pxc;
  const int tile = blockIdx.y * gridDim.x + blockIdx.x, ntiles = gridDim.x * gridDim.y;
  const int npx = min(pxc, OW - x0);
  const int qpp = S >> 2, planes = 256 / qpp;
  const int plane = tid / qpp, q = tid % qpp;
  const int px0 = plane * LAY_EPT, q4 = q * 4;
  float ty[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) ty[r] = lin01(min((int)(((int64_t)min(y0 + r, OH - 1) * H) / OH), H - 1), H);
  // this thread's 8 pixels x ROWS rows of dout, one channel quad
  float4 d[ROWS][LAY_EPT];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < LAY_EPT; ++i) {
      d[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (y0 + r < OH && px0 + i < npx)
        d[r][i] = *(const float4*)(dout + ((int64_t)(b * OH + y0 + r) * OW + x0 + px0 + i) * out_cs + out_off + q4);
    }
  const float* bx = boxes + (int64_t)b * O * 4;
  const uint8_t* vb = valid + (int64_t)b * O;
  uint8_t* fl = flags + ((int64_t)b * ntiles + tile) * O;
  float* pt = partial + ((int64_t)b * ntiles + tile) * O * S;
  const int xs0 = min((int)(((int64_t)x0 * W) / OW), W - 1);
  const int xs1 = min((int)(((int64_t)(x0 + npx - 1) * W) / OW), W - 1);
  const float step = W > 1 ? 1.0f / (float)(W - 1) : 1.0f;
  const float tx_lo = lin01(xs0, W) - step, tx_hi = lin01(xs1, W) + step;

  for (int ob = 0; ob < O; ob += LAY_CULL) {
    __syncthreads();
    {
      const int o = ob + tid;
      float wy[ROWS];
      bool act = false;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) wy[r] = 0.f;
      if (o < O && vb[o]) {
        const float4 bq = *(const float4*)(bx + o * 4);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          wy[r] = (y0 + r < OH) ? coverage(ty[r], bq.y, bq.w) : 0.f;
          act = act || (wy[r] != 0.0f);
        }
        if (act) {                          // the forward's x-support test: a dropped object has zero weight in the tile
          const float lo = fminf(bq.x - bq.z / 16.0f, bq.x + bq.z * (1.0f + 1.0f / 16.0f));
          const float hi = fmaxf(bq.x - bq.z / 16.0f, bq.x + bq.z * (1.0f + 1.0f / 16.0f));
          if (tx_hi < lo || tx_lo > hi) act = false;
        }
      }
      if (o < O) fl[o] = act ? 1 : 0;
      const unsigned long long m = __ballot(act);
      const int lane = tid & 63, wv = tid >> 6;
      if (lane == 0) s_cnt[wv] = __popcll(m);
      __syncthreads();
      int base = 0;
      for (int w = 0; w < wv; ++w) base += s_cnt[w];
      const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
      if (act) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) s_wy[slot * ROWS + r] = wy[r];
        s_act[slot] = o;
      }
    }
    __syncthreads();
    const int ntot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    for (int a0 = 0; a0 < ntot; a0 += LAY_BB) {
      const int nact = min(LAY_BB, ntot - a0);
      if (a0 > 0) __syncthreads();
      for (int i = tid; i < nact * pxc; i += 256) {
        const int a = i / pxc, xl = i - a * pxc;
        const int o = s_act[a0 + a];
        const int xsrc = min((int)(((int64_t)min(x0 + xl, OW - 1) * W) / OW), W - 1);
        s_wx[a * pxc + xl] = xl < npx ? coverage(lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]) : 0.f;
      }
      __syncthreads();
      for (int a = 0; a < nact; ++a) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // planes beyond the chunk (pxc < planes * LAY_EPT on narrow pyramid levels) hold no pixels: they must not read
        // s_wx past row a (unwritten LDS: 0 * NaN would poison the sum) and contribute an exact zero
        if (px0 < npx) {
          const float4 w0 = *(const float4*)&s_wx[a * pxc + px0], w1 = *(const float4*)&s_wx[a * pxc + px0 + 4];
          const float wv[LAY_EPT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float wy = s_wy[(a0 + a) * ROWS + r];
#pragma unroll
            for (int i = 0; i < LAY_EPT; ++i) {
              const float w = wy * wv[i];
              acc.x += d[r][i].x * w; acc.y += d[r][i].y * w; acc.z += d[r][i].z * w; acc.w += d[r][i].w * w;
            }
          }
        }
        s_part[a * 256 + plane * qpp + q] = acc;
      }
      __syncthreads();
      for (int e = tid; e < nact * qpp; e += 256) {   // sum over the pixel planes in plane order
        const int a = e / qpp, qq = e - a * qpp;
        float4 t = s_part[a * 256 + qq];
        for (int pl = 1; pl < planes; ++pl) {
          const float4 v = s_part[a * 256 + pl * qpp + qq];
          t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *(float4*)(pt + (int64_t)s_act[a0 + a] * S + qq * 4) = t;
      }
    }
  }
}

// pass 2: dvecs[b][o] = (accumulate ? dvecs : 0) + sum over the tiles that flagged o, in tile order
__global__ __launch_bounds__(256) void k_layout_bwd_gather(const float* __restrict__ partial,
                                                            const uint8_t* __restrict__ flags,
                                                            const uint8_t* __restrict__ valid, int O, int S, int ntiles,
                                                            float* __restrict__ dvecs, int accumulate) {
  __shared__ float4 s_red[256];
  const int tid = threadIdx.x, o = blockIdx.x, b = blockIdx.y;
  const int qpp = S >> 2, lanes = 256 / qpp;
  const int q = tid % qpp, tl = tid / qpp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid[(int64_t)b * O + o]) {
    for (int t = tl; t < ntiles; t += lanes) {
      const int64_t bt = (int64_t)b * ntiles + t;
      if (flags[bt * O + o]) {
        const float4 v = *(const float4*)(partial + (bt * O + o) * S + q * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
  }
  s_red[tid] = acc;
  __syncthreads();
  if (tl == 0) {
    for (int l = 1; l < lanes; ++l) {
      const float4 v = s_red[l * qpp + q];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float* dv = dvecs + ((int64_t)b * O + o) * S + q * 4;
    if (accumulate) {
      const float4 old = *(const float4*)dv;
      acc.x += old.x; acc.y += old.y; acc.z += old.z; acc.w += old.w;
    }
    *(float4*)dv = acc;
  }
}

// One block per (object, image): reduce dout over the box's support only.  256 threads for small layouts,
// 1024 for >= 64x64 ones (S/4 lanes per pixel, blockDim/(S/4) pixels in flight, two loads per lane in flight).
__global__ __launch_bounds__(1024) void k_layout_bwd(const float* __restrict__ dout, int out_cs, int out_off,
                                                     const float* __restrict__ boxes,
                                                     const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ masks, int M, int O, int S, int H,
                                                     int W, int OH, int OW, float* __restrict__ dvecs,
                                                     int accumulate, const float* __restrict__ vecs,
                                                     float* __restrict__ dboxes) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wy = sm;            // [OH]
  float* s_wx = sm + OH;       // [OW]
  int* s_rng = (int*)(s_wx + OW);  // ylo, yhi, xlo, xhi
  float* s_red = sm + (((OH + OW + 4) + 3) & ~3);  // [max(npl * S, BD * 4)], 16-byte aligned
  float* s_dwy = s_red + max((blockDim.x / (S >> 2)) * S, blockDim.x * 4);   // [OH]  box gradients only
  float* s_dwx = s_dwy + OH;                                                    // [OW]
  const int tid = threadIdx.x, BD = blockDim.x;
  const int o = blockIdx.x, b = blockIdx.y;
  float* dv = dvecs + ((int64_t)b * O + o) * S;
  float* db = dboxes != nullptr ? dboxes + ((int64_t)b * O + o) * 4 : nullptr;
  if (!valid[(int64_t)b * O + o]) {
    if (!accumulate) {
      for (int d = tid; d < S; d += BD) dv[d] = 0.f;
      if (db != nullptr && tid < 4) db[tid] = 0.f;
    }
    return;
  }
  const float* bx = boxes + ((int64_t)b * O + o) * 4;
  const float x0 = bx[0], y0 = bx[1], ww = bx[2], hh = bx[3];
  if (tid < 4) s_rng[tid] = (tid & 1) ? -1 : (1 << 30);
  __syncthreads();
  for (int y = tid; y < OH; y += BD) {
    int ysrc = min((int)(((int64_t)y * H) / OH), H - 1);
    float w = masks == nullptr ? coverage(lin01(ysrc, H), y0, hh) : coverage_n(lin01(ysrc, H), y0, hh, M);
    s_wy[y] = w;
    bool in = w != 0.f;
    if (db != nullptr) {                     // the derivative's support includes the (measure-zero) points where
      const float dw = coverage_dix(lin01(ysrc, H), y0, hh, masks == nullptr ? 8 : M);     // the coverage itself is 0
      s_dwy[y] = dw;
      in = in || dw != 0.f;
    }
    if (in) {
      atomicMin(&s_rng[0], y);
      atomicMax(&s_rng[1], y);
    }
  }
  for (int x = tid; x < OW; x += BD) {
    int xsrc = min((int)(((int64_t)x * W) / OW), W - 1);
    float w = masks == nullptr ? coverage(lin01(xsrc, W), x0, ww) : coverage_n(lin01(xsrc, W), x0, ww, M);
    s_wx[x] = w;
    bool in = w != 0.f;
    if (db != nullptr) {
      const float dw = coverage_dix(lin01(xsrc, W), x0, ww, masks == nullptr ? 8 : M);
      s_dwx[x] = dw;
      in = in || dw != 0.f;
    }
    if (in) {
      atomicMin(&s_rng[2], x);
      atomicMax(&s_rng[3], x);
    }
  }
  __syncthreads();
  const int ylo = s_rng[0], yhi = s_rng[1], xlo = s_rng[2], xhi = s_rng[3];
  const int qpp = S >> 2;
  const int npl = BD / qpp;
  const int q = tid % qpp, pl = tid / qpp;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (yhi >= ylo && xhi >= xlo && pl < npl) {
    const int nx = xhi - xlo + 1, npix = (yhi - ylo + 1) * nx;
    const float* base = dout + (int64_t)b * OH * OW * out_cs + out_off + q * 4;
    auto weight = [&](int yy, int xx) -> float {
      if (masks == nullptr) return s_wy[yy] * s_wx[xx];
      int iy0;
      float wy0, wy1;
      axis_taps(lin01(min((int)(((int64_t)yy * H) / OH), H - 1), H), y0, hh, M, iy0, wy0, wy1);
      return mask_weight(masks + ((int64_t)b * O + o) * M * M, M, iy0, wy0, wy1,
                         lin01(min((int)(((int64_t)xx * W) / OW), W - 1), W), x0, ww);
    };
    float4 acc1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int i = pl;
    for (; i + npl < npix; i += 2 * npl) {            // two independent loads in flight per lane
      const int ya = i / nx, xa = xlo + (i - ya * nx);
      const int j = i + npl;
      const int yb = j / nx, xb = xlo + (j - yb * nx);
      const float4 ga = *(const float4*)&base[((int64_t)(ya + ylo) * OW + xa) * out_cs];
      const float4 gb = *(const float4*)&base[((int64_t)(yb + ylo) * OW + xb) * out_cs];
      const float wa = weight(ya + ylo, xa), wb = weight(yb + ylo, xb);
      acc.x += ga.x * wa; acc.y += ga.y * wa; acc.z += ga.z * wa; acc.w += ga.w * wa;
      acc1.x += gb.x * wb; acc1.y += gb.y * wb; acc1.z += gb.z * wb; acc1.w += gb.w * wb;
    }
    if (i < npix) {
      const int ya = i / nx, xa = xlo + (i - ya * nx);
      const float4 ga = *(const float4*)&base[((int64_t)(ya + ylo) * OW + xa) * out_cs];
      const float wa = weight(ya + ylo, xa);
      acc.x += ga.x * wa; acc.y += ga.y * wa; acc.z += ga.z * wa; acc.w += ga.w * wa;
    }
    acc.x += acc1.x; acc.y += acc1.y; acc.z += acc1.z; acc.w += acc1.w;
  }
  if (pl < npl) *(float4*)&s_red[pl * S + q * 4] = acc;
  __syncthreads();
  for (int d = tid; d < S; d += BD) {
    float t = 0.f;
    for (int p = 0; p < npl; ++p) t += s_red[p * S + d];
    dv[d] = accumulate ? dv[d] + t : t;
  }
  if (db == nullptr) return;
  // ---- gradient w.r.t. the box (reference layout.py:98-110 is differentiable in x0, y0, w, h): with
  // G(y,x) = sum_d dout[d,y,x] * vec[d] and ix = n (t - lo) / size - 1/2 (n = 8 or M),
  //   d/dx0 = -n/ww   * sum G * dweight/dix,          d/dww = -n/ww^2 * sum G * dweight/dix * (tx - x0)   (same in y)
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);       // A1, A2 (x), B1, B2 (y)
  if (yhi >= ylo && xhi >= xlo && pl < npl) {
    const int nx = xhi - xlo + 1, npix = (yhi - ylo + 1) * nx;
    const float* base = dout + (int64_t)b * OH * OW * out_cs + out_off + q * 4;
    const float4 v4 = *(const float4*)&vecs[((int64_t)b * O + o) * S + q * 4];
    for (int i = pl; i < npix; i += npl) {
      const int yy = i / nx + ylo, xx = xlo + (i - (i / nx) * nx);
      const float ty = lin01(min((int)(((int64_t)yy * H) / OH), H - 1), H);
      const float tx = lin01(min((int)(((int64_t)xx * W) / OW), W - 1), W);
      float cx, cy;                                   // dweight/dix, dweight/diy at this pixel
      if (masks == nullptr) {
        cx = s_wy[yy] * s_dwx[xx];
        cy = s_dwy[yy] * s_wx[xx];
      } else {
        float w;
        mask_weight_grad(masks + ((int64_t)b * O + o) * M * M, M, ty, y0, hh, tx, x0, ww, w, cx, cy);
      }
      if (cx == 0.f && cy == 0.f) continue;
      const float4 ga = *(const float4*)&base[((int64_t)yy * OW + xx) * out_cs];
      const float g = ga.x * v4.x + ga.y * v4.y + ga.z * v4.z + ga.w * v4.w;
      bsum.x += g * cx;
      bsum.y += g * cx * (tx - x0);
      bsum.z += g * cy;
      bsum.w += g * cy * (ty - y0);
    }
  }
  __syncthreads();                                     // the dvecs reduction is done with s_red
  *(float4*)&s_red[tid * 4] = bsum;
  __syncthreads();
  if (tid < 4) {
    float t = 0.f;
    for (int p = 0; p < BD; ++p) t += s_red[p * 4 + tid];       // fixed order
    const float n = masks == nullptr ? 8.0f : (float)M;
    const float size = (tid < 2) ? ww : hh;
    const float sc = (tid & 1) ? -n / (size * size) : -n / size;
    // (A1, A2, B1, B2) -> (dx0, dww, dy0, dhh); boxes are stored [x0, y0, w, h]
    const int slot = tid == 0 ? 0 : (tid == 1 ? 2 : (tid == 2 ? 1 : 3));
    db[slot] = accumulate ? db[slot] + t * sc : t * sc;
  }
}

// Gradient w.r.t. the masks (masks_to_layout, layout.py:48-77: grid_sample is differentiable in its input):
//   dmask[m][n] = sum_{y,x} G(y,x) * Wy(y,m) * Wx(x,n),   G(y,x) = sum_d dout[d,y,x] * vec[d],
// Wy(y,m) the bilinear weight output row y puts on mask row m (at most two rows per y).  One block per (object, image),
// one thread per mask cell at a time; every cell sums its pixels in ascending (y, x) order: bit-reproducible.  G is
// recomputed by the (up to four) cells a pixel touches — this pass exists for callers that train through predicted
// masks (model.py's mask_net), not for the benchmarked GT-mask configurations.
__global__ __launch_bounds__(256) void k_layout_bwd_masks(const float* __restrict__ dout, int out_cs, int out_off,
                                                           const float* __restrict__ boxes,
                                                           const uint8_t* __restrict__ valid, int M, int O, int S, int H,
                                                           int W, int OH, int OW, const float* __restrict__ vecs,
                                                           float* __restrict__ dmasks, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wy0 = sm;                 // [OH]
  float* s_wy1 = s_wy0 + OH;         // [OH]
  float* s_wx0 = s_wy1 + OH;         // [OW]
  float* s_wx1 = s_wx0 + OW;         // [OW]
  int* s_iy = (int*)(s_wx1 + OW);    // [OH]
  int* s_ix = s_iy + OH;             // [OW]
  float* s_vec = (float*)(s_ix + OW);  // [S]
  const int tid = threadIdx.x;
  const int o = blockIdx.x, b = blockIdx.y;
  float* dm = dmasks + ((int64_t)b * O + o) * M * M;
  if (!valid[(int64_t)b * O + o]) {
    if (!accumulate)
      for (int c = tid; c < M * M; c += 256) dm[c] = 0.f;
    return;
  }
  const float* bx = boxes + ((int64_t)b * O + o) * 4;
  const float x0 = bx[0], y0 = bx[1], ww = bx[2], hh = bx[3];
  for (int y = tid; y < OH; y += 256) {
    const int ysrc = min((int)(((int64_t)y * H) / OH), H - 1);
    int i0;
    float w0, w1;
    axis_taps(lin01(ysrc, H), y0, hh, M, i0, w0, w1);
    s_iy[y] = i0; s_wy0[y] = w0; s_wy1[y] = w1;
  }
  for (int x = tid; x < OW; x += 256) {
    const int xsrc = min((int)(((int64_t)x * W) / OW), W - 1);
    int i0;
    float w0, w1;
    axis_taps(lin01(xsrc, W), x0, ww, M, i0, w0, w1);
    s_ix[x] = i0; s_wx0[x] = w0; s_wx1[x] = w1;
  }
  for (int d = tid; d < S; d += 256) s_vec[d] = vecs[((int64_t)b * O + o) * S + d];
  __syncthreads();
  const float* base = dout + (int64_t)b * OH * OW * out_cs + out_off;
  for (int c = tid; c < M * M; c += 256) {
    const int m = c / M, n = c - m * M;
    float acc = 0.f;
    for (int y = 0; y < OH; ++y) {
      const int iy = s_iy[y];
      const float wy = iy == m ? s_wy0[y] : (iy + 1 == m ? s_wy1[y] : 0.f);
      if (wy == 0.f) continue;
      float row = 0.f;
      for (int x = 0; x < OW; ++x) {
        const int ix = s_ix[x];
        const float wx = ix == n ? s_wx0[x] : (ix + 1 == n ? s_wx1[x] : 0.f);
        if (wx == 0.f) continue;
        const float4* g4 = (const float4*)&base[((int64_t)y * OW + x) * out_cs];
        float g = 0.f;
        for (int q = 0; q < (S >> 2); ++q) {
          const float4 v = g4[q];
          g += v.x * s_vec[4 * q] + v.y * s_vec[4 * q + 1] + v.z * s_vec[4 * q + 2] + v.w * s_vec[4 * q + 3];
        }
        row += g * wx;
      }
      acc += row * wy;
    }
    dm[c] = accumulate ? dm[c] + acc : acc;
  }
}

// ---------------------------------------------------------------------------------- test mode
// masks_to_layout(test_mode=True) (reference sg2im/layout.py:71-74,135-151): objects are painted in ascending
// order of their "mass" sum(samples[j]); a pixel belongs to the FIRST object in that order whose bilinearly
// sampled mask exceeds 0.5 there, and receives that object's sample vec[j] * mask_sample — one object per pixel.

// mass[b,o] = sum_{d,y,x} vec[d] * mask_sample(y,x) at full resolution; one block per (object, image)
__global__ __launch_bounds__(256) void k_layout_mass(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                      const uint8_t* __restrict__ valid,
                                                      const float* __restrict__ masks, int M, int O, int S, int H,
                                                      int W, float* __restrict__ mass) {
  __shared__ float s_red[256];
  const int tid = threadIdx.x, o = blockIdx.x, b = blockIdx.y;
  float acc = 0.f;
  const bool ok = valid[(int64_t)b * O + o] != 0;
  if (ok) {
    const float* bx = boxes + ((int64_t)b * O + o) * 4;
    const float x0 = bx[0], y0 = bx[1], ww = bx[2], hh = bx[3];
    const float* mk = masks + ((int64_t)b * O + o) * M * M;
    for (int i = tid; i < H * W; i += 256) {
      const int y = i / W, x = i - y * W;
      int iy0;
      float wy0, wy1;
      axis_taps(lin01(y, H), y0, hh, M, iy0, wy0, wy1);
      if (wy0 == 0.f && wy1 == 0.f) continue;
      acc += mask_weight(mk, M, iy0, wy0, wy1, lin01(x, W), x0, ww);
    }
  }
  s_red[tid] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) s_red[tid] += s_red[tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    float vs = 0.f;
    if (ok)
      for (int d = 0; d < S; ++d) vs += vecs[((int64_t)b * O + o) * S + d];
    mass[(int64_t)b * O + o] = ok ? vs * s_red[0] : __builtin_inff();
  }
}

// one thread per output pixel finds the owner; the block then writes the S channels cooperatively
__global__ __launch_bounds__(256) void k_layout_paint(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                       const float* __restrict__ masks, int M,
                                                       const int* __restrict__ order, int O, int S, int H, int W, int OH,
                                                       int OW, float* __restrict__ out, int out_cs, int out_off) {
  __shared__ int s_own[256];
  __shared__ float s_w[256];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, y = blockIdx.y, xb = blockIdx.x * 256;
  const int npx = min(256, OW - xb);
  const int ysrc = min((int)(((int64_t)y * H) / OH), H - 1);
  const float ty = lin01(ysrc, H);
  int own = -1;
  float wv = 0.f;
  if (tid < npx) {
    const int xsrc = min((int)(((int64_t)(xb + tid) * W) / OW), W - 1);
    const float tx = lin01(xsrc, W);
    for (int k = 0; k < O; ++k) {
      const int o = order[(int64_t)b * O + k];
      if (o < 0) break;
      const float* bx = boxes + ((int64_t)b * O + o) * 4;
      int iy0;
      float wy0, wy1;
      axis_taps(ty, bx[1], bx[3], M, iy0, wy0, wy1);
      if (wy0 == 0.f && wy1 == 0.f) continue;
      const float c = mask_weight(masks + ((int64_t)b * O + o) * M * M, M, iy0, wy0, wy1, tx, bx[0], bx[2]);
      if (c > 0.5f) {
        own = o;
        wv = c;
        break;
      }
    }
  }
  s_own[tid] = own;
  s_w[tid] = wv;
  __syncthreads();
  const int qpp = S >> 2;
  float* orow = out + ((int64_t)(b * OH + y) * OW + xb) * out_cs + out_off;
  for (int e = tid; e < npx * qpp; e += 256) {
    const int px = e / qpp, q = e - px * qpp;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int o = s_own[px];
    if (o >= 0) {
      const float4 u = *(const float4*)&vecs[((int64_t)b * O + o) * S + q * 4];
      const float w = s_w[px];
      v = make_float4(u.x * w, u.y * w, u.z * w, u.w * w);
    }
    *(float4*)&orow[(int64_t)px * out_cs + q * 4] = v;
  }
}

extern "C" {

int csg_layout_mass(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                    int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, float* mass, void* stream) {
  CSG_REQUIRE(masks != nullptr && M >= 1 && M <= 1024, CSG_E_BADSHAPE, "csg_layout_mass: bad mask size %ld", (long)M);
  CSG_REQUIRE(B > 0 && B <= 65535 && O >= 0 && S > 0 && H > 0 && W > 0, CSG_E_BADSHAPE, "csg_layout_mass: bad shape");
  if (O == 0) return CSG_OK;
  CSG_LAUNCH(k_layout_mass, dim3((unsigned)O, (unsigned)B), dim3(256), 0, (hipStream_t)stream, vecs, boxes, valid,
                     masks, (int)M, (int)O, (int)S, (int)H, (int)W, mass);
  return check_launch("csg_layout_mass");
}

int csg_layout_paint(const float* vecs, const float* boxes, const float* masks, int64_t M, const int32_t* order,
                     int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                     int64_t out_cs, int64_t out_off, void* stream) {
  CSG_REQUIRE(masks != nullptr && M >= 1 && M <= 1024, CSG_E_BADSHAPE, "csg_layout_paint: bad mask size %ld", (long)M);
  CSG_REQUIRE(B > 0 && O >= 0 && S > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, CSG_E_BADSHAPE,
              "csg_layout_paint: bad shape");
  CSG_REQUIRE(S % 4 == 0 && out_cs % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)out % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_layout_paint: S, out_cs, out_off must be multiples of 4 (16-byte rows)");
  CSG_REQUIRE(OH <= 65535 && B <= 65535, CSG_E_UNSUPPORTED, "csg_layout_paint: OH/B too large");
  CSG_LAUNCH(k_layout_paint, dim3((unsigned)cdiv(OW, 256), (unsigned)OH, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, vecs, boxes, masks, (int)M, order, (int)O, (int)S, (int)H, (int)W, (int)OH,
                     (int)OW, out, (int)out_cs, (int)out_off);
  return check_launch("csg_layout_paint");
}

static int layout_fwd_launch(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                             int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                             int64_t out_cs, int64_t out_off, const LayTail& tail, void* stream);

int csg_layout_fwd(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                   int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                   int64_t out_cs, int64_t out_off, void* stream) {
  const LayTail none = {nullptr, 0, 0, 0, 0, 0};
  return layout_fwd_launch(vecs, boxes, valid, masks, M, B, O, S, H, W, OH, OW, out, out_cs, out_off, none, stream);
}

int csg_disc_input_fwd(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M, int64_t B,
                       int64_t O, int64_t S, int64_t H, int64_t W, const float* img, int64_t img_sb, int64_t img_sc,
                       int64_t img_sh, int64_t img_sw, float* out, int64_t out_cs, void* stream) {
  CSG_REQUIRE(img != nullptr && out_cs >= S + 4 && out_cs % 4 == 0, CSG_E_BADSHAPE,
              "csg_disc_input_fwd: needs an image and a pixel stride of at least S + 4 = %ld floats (got %ld)", (long)(S + 4),
              (long)out_cs);
  const LayTail tail = {img, (long long)img_sb, (long long)img_sc, (long long)img_sh, (long long)img_sw, (int)((out_cs - S) / 4)};
  return layout_fwd_launch(vecs, boxes, valid, masks, M, B, O, S, H, W, H, W, out, out_cs, 0, tail, stream);
}

static int layout_fwd_launch(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                             int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                             int64_t out_cs, int64_t out_off, const LayTail& tail, void* stream) {
  CSG_REQUIRE(masks == nullptr || (M >= 1 && M <= 1024), CSG_E_BADSHAPE, "csg_layout_fwd: bad mask size %ld", (long)M);
  CSG_REQUIRE(B > 0 && O >= 0 && S > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, CSG_E_BADSHAPE,
              "csg_layout_fwd: bad shape");
  CSG_REQUIRE(S % 4 == 0 && out_cs % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)out % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_layout_fwd: S=%ld, out_cs=%ld, out_off=%ld must be multiples of 4 (16-byte rows)", (long)S,
              (long)out_cs, (long)out_off);
  CSG_REQUIRE(S <= 1024 && OH <= 65535 && B <= 65535, CSG_E_UNSUPPORTED, "csg_layout_fwd: S/OH/B too large");
  hipStream_t s = (hipStream_t)stream;
  const int qpp = (int)(S / 4);
  int pxc = (LAY_EPT * 256) / qpp;
  if (pxc > LAY_PXC) pxc = LAY_PXC;
  if (pxc > OW) pxc = (int)OW;
  CSG_REQUIRE(pxc >= 1, CSG_E_UNSUPPORTED, "csg_layout_fwd: S too large for one chunk");
  size_t shm = (size_t)(LAY_OB * pxc + LAY_OB * S) * 4 + (size_t)4 * LAY_CULL * 4 + 16;
  ProfScope p(K_LAYOUT_FWD, (double)B * OH * OW * S * 4, s);  // algorithmic bytes: the output, once
  if (masks == nullptr && (256 % qpp) == 0 && (pxc % LAY_EPT) == 0 && OH >= 32) {
    // boxes_to_layout on maps from 32 rows up: four rows per block share the staged x coverage and object vectors
    constexpr int ROWS = 4;
    const size_t shm4 = (size_t)(LAY_OB * pxc + LAY_OB * S) * 4 + (size_t)LAY_CULL * ROWS * 4 + (size_t)LAY_CULL * 4 + 16;
    dim3 grid4((unsigned)cdiv(OW, pxc), (unsigned)cdiv(OH, ROWS), (unsigned)B);
    CSG_LAUNCH(k_layout_fwd_rows<ROWS>, grid4, dim3(256), shm4, s, vecs, boxes, valid, (int)O, (int)S, (int)H, (int)W,
                       (int)OH, (int)OW, pxc, out, (int)out_cs, (int)out_off, tail);
    return check_launch("csg_layout_fwd");
  }
  dim3 grid((unsigned)cdiv(OW, pxc), (unsigned)OH, (unsigned)B);
  CSG_LAUNCH(k_layout_fwd, grid, dim3(256), shm, s, vecs, boxes, valid, masks, (int)M, (int)O, (int)S, (int)H,
                     (int)W, (int)OH, (int)OW, pxc, out, (int)out_cs, (int)out_off, tail);
  return check_launch("csg_layout_fwd");
}

// tiled two-pass backward: boxes_to_layout, no box gradients, S/4 dividing 256, maps from 32 rows up
static bool layout_bwd_tiled(const float* masks, const float* dboxes, int64_t S, int64_t OH, int64_t OW, int64_t O, int& pxc,
                             int& ntiles) {
  if (masks != nullptr || dboxes != nullptr || S % 4 != 0 || O <= 0) return false;
  const int qpp = (int)(S / 4);
  if (qpp > 256 || (256 % qpp) != 0 || OH < 32) return false;
  pxc = (LAY_EPT * 256) / qpp;
  if (pxc > LAY_PXC) pxc = LAY_PXC;
  if (pxc > OW) pxc = (int)OW;
  if (pxc % LAY_EPT) return false;
  ntiles = (int)(cdiv(OW, pxc) * cdiv(OH, 4));
  return true;
}

int64_t csg_layout_bwd_workspace(int64_t B, int64_t O, int64_t S, int64_t OH, int64_t OW, int32_t has_masks,
                                 int32_t box_gradients) {
  int pxc = 0, ntiles = 0;
  if (B <= 0 || !layout_bwd_tiled(has_masks ? (const float*)1 : nullptr, box_gradients ? (const float*)1 : nullptr, S, OH,
                                  OW, O, pxc, ntiles))
    return 0;
  const int64_t part = B * ntiles * O * S * 4;
  const int64_t flags = (B * ntiles * O + 15) / 16 * 16;
  return part + flags;
}

int csg_layout_bwd_masks(const float* dout, int64_t out_cs, int64_t out_off, const float* boxes, const uint8_t* valid,
                         int64_t M, int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW,
                         const float* vecs, float* dmasks, int accumulate, void* stream) {
  CSG_REQUIRE(M >= 1 && M <= 1024, CSG_E_BADSHAPE, "csg_layout_bwd_masks: bad mask size %ld", (long)M);
  CSG_REQUIRE(B > 0 && O >= 0 && S > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, CSG_E_BADSHAPE,
              "csg_layout_bwd_masks: bad shape");
  CSG_REQUIRE(S % 4 == 0 && out_cs % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)dout % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_layout_bwd_masks: S, out_cs, out_off must be multiples of 4");
  CSG_REQUIRE(S <= 1024 && OH <= 4096 && OW <= 4096 && B <= 65535, CSG_E_UNSUPPORTED, "csg_layout_bwd_masks: too large");
  if (O == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  const size_t shm = (size_t)(3 * (OH + OW) + S) * 4;
  ProfScope p(K_LAYOUT_BWD, (double)B * OH * OW * S * 4, s);
  CSG_LAUNCH(k_layout_bwd_masks, dim3((unsigned)O, (unsigned)B), dim3(256), shm, s, dout, (int)out_cs, (int)out_off, boxes,
             valid, (int)M, (int)O, (int)S, (int)H, (int)W, (int)OH, (int)OW, vecs, dmasks, accumulate);
  return check_launch("csg_layout_bwd_masks");
}

int csg_layout_bwd(const float* dout, int64_t out_cs, int64_t out_off, const float* boxes, const uint8_t* valid,
                   const float* masks, int64_t M, int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH,
                   int64_t OW, float* dvecs, int accumulate, const float* vecs, float* dboxes, void* workspace,
                   int64_t workspace_bytes, void* stream) {
  {
    int pxc = 0, ntiles = 0;
    if (B > 0 && workspace != nullptr && layout_bwd_tiled(masks, dboxes, S, OH, OW, O, pxc, ntiles) &&
        workspace_bytes >= csg_layout_bwd_workspace(B, O, S, OH, OW, 0, 0) && B <= 65535 && OH <= 4 * 65535) {
      CSG_REQUIRE(out_cs % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)workspace % 16) == 0,
                  CSG_E_UNSUPPORTED, "csg_layout_bwd: out_cs, out_off must be multiples of 4, pointers 16-byte aligned");
      constexpr int ROWS = 4;
      hipStream_t s = (hipStream_t)stream;
      float* partial = (float*)workspace;
      uint8_t* flags = (uint8_t*)workspace + B * ntiles * O * S * 4;
      const size_t shm = (size_t)LAY_BB * 256 * 16 + (size_t)LAY_BB * pxc * 4 + (size_t)LAY_CULL * ROWS * 4 +
                         (size_t)LAY_CULL * 4 + 16;
      ProfScope p(K_LAYOUT_BWD, (double)B * OH * OW * S * 4, s);
      dim3 grid((unsigned)cdiv(OW, pxc), (unsigned)cdiv(OH, ROWS), (unsigned)B);
      CSG_LAUNCH(k_layout_bwd_tiles<ROWS>, grid, dim3(256), shm, s, dout, (int)out_cs, (int)out_off, boxes, valid,
                         (int)O, (int)S, (int)H, (int)W, (int)OH, (int)OW, pxc, partial, flags);
      int rc = check_launch("csg_layout_bwd(tiles)");
      if (rc) return rc;
      CSG_LAUNCH(k_layout_bwd_gather, dim3((unsigned)O, (unsigned)B), dim3(256), 0, s, partial, flags, valid, (int)O,
                         (int)S, ntiles, dvecs, accumulate);
      return check_launch("csg_layout_bwd(gather)");
    }
  }
  CSG_REQUIRE(masks == nullptr || (M >= 1 && M <= 1024), CSG_E_BADSHAPE, "csg_layout_bwd: bad mask size %ld", (long)M);
  CSG_REQUIRE(B > 0 && O >= 0 && S > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, CSG_E_BADSHAPE,
              "csg_layout_bwd: bad shape");
  CSG_REQUIRE(S % 4 == 0 && out_cs % 4 == 0 && out_off % 4 == 0 && ((uintptr_t)dout % 16) == 0, CSG_E_UNSUPPORTED,
              "csg_layout_bwd: S, out_cs, out_off must be multiples of 4");
  CSG_REQUIRE(S <= 1024 && OH <= 4096 && OW <= 4096 && B <= 65535, CSG_E_UNSUPPORTED, "csg_layout_bwd: too large");
  if (O == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  const int qpp = (int)(S / 4);
  const int bd = (OH * OW >= 64 * 64) ? 1024 : 256;
  const int npl = bd / qpp;
  CSG_REQUIRE(npl >= 1, CSG_E_UNSUPPORTED, "csg_layout_bwd: S too large");
  CSG_REQUIRE(dboxes == nullptr || vecs != nullptr, CSG_E_BADSHAPE, "csg_layout_bwd: box gradients need vecs");
  const size_t red = (size_t)npl * S > (size_t)bd * 4 ? (size_t)npl * S : (size_t)bd * 4;
  size_t shm = (size_t)(((OH + OW + 4) + 3) & ~3) * 4 + red * 4 + (dboxes != nullptr ? (size_t)(OH + OW) * 4 : 0);
  ProfScope p(K_LAYOUT_BWD, (double)B * OH * OW * S * 4, s);
  CSG_LAUNCH(k_layout_bwd, dim3((unsigned)O, (unsigned)B), dim3((unsigned)bd), shm, s, dout, (int)out_cs, (int)out_off,
                     boxes, valid, masks, (int)M, (int)O, (int)S, (int)H, (int)W, (int)OH, (int)OW, dvecs, accumulate,
                     vecs, dboxes);
  return check_launch("csg_layout_bwd");
}

}  // extern "C"
