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

#define LAY_OB 32    // objects per LDS batch
#define LAY_PXC 256  // max pixels per block chunk
#define LAY_EPT 8    // float4 elements per thread

__global__ __launch_bounds__(256) void k_layout_fwd(const float* __restrict__ vecs, const float* __restrict__ boxes,
                                                     const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ masks, int M, int O, int S, int H,
                                                     int W, int OH, int OW, int pxc, float* __restrict__ out,
                                                     int out_cs, int out_off) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wx = sm;                       // [LAY_OB][pxc]
  float* s_vec = sm + LAY_OB * pxc;       // [LAY_OB][S]
  float* s_wy = s_vec + LAY_OB * S;       // [LAY_OB]
  int* s_act = (int*)(s_wy + LAY_OB);     // [LAY_OB] object index of each active slot
  int* s_n = s_act + LAY_OB;              // [1]
  float* s_wy1 = (float*)(s_n + 4);       // [LAY_OB] mask path: weight of the upper row tap
  int* s_iy0 = (int*)(s_wy1 + LAY_OB);    // [LAY_OB] mask path: lower row tap index

  const int tid = threadIdx.x;
  const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * pxc;
  const int npx = min(pxc, OW - x0);
  const int qpp = S >> 2;
  const int nel = npx * qpp;
  const int ysrc = min((int)(((int64_t)y * H) / OH), H - 1);
  const float ty = lin01(ysrc, H);

  float4 acc[LAY_EPT];
  int eq[LAY_EPT], ex[LAY_EPT];
#pragma unroll
  for (int i = 0; i < LAY_EPT; ++i) {
    acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = tid + 256 * i;
    ex[i] = e / qpp;
    eq[i] = e - ex[i] * qpp;
  }
  const float* bx = boxes + (int64_t)b * O * 4;
  const uint8_t* vb = valid + (int64_t)b * O;
  const float* vv = vecs + (int64_t)b * O * S;

  for (int ob = 0; ob < O; ob += LAY_OB) {
    __syncthreads();
    if (tid < 64) {  // wave 0: row coverage + ordered compaction of the active objects
      int o = ob + tid;
      float wy = 0.f;
      bool act = false;
      float wy1 = 0.f;
      int iy0 = 0;
      if (tid < LAY_OB && o < O && vb[o]) {
        if (masks == nullptr) {
          wy = coverage(ty, bx[o * 4 + 1], bx[o * 4 + 3]);
          act = (wy != 0.0f);
        } else {
          axis_taps(ty, bx[o * 4 + 1], bx[o * 4 + 3], M, iy0, wy, wy1);
          act = (wy != 0.0f) || (wy1 != 0.0f);
        }
      }
      unsigned long long m = __ballot(act);
      int slot = __popcll(m & ((1ull << tid) - 1ull));
      if (act) {
        s_wy[slot] = wy;
        s_wy1[slot] = wy1;
        s_iy0[slot] = iy0;
        s_act[slot] = o;
      }
      if (tid == 0) *s_n = __popcll(m);
    }
    __syncthreads();
    const int nact = *s_n;
    if (nact == 0) continue;
    for (int i = tid; i < nact * npx; i += 256) {
      int a = i / npx, xl = i - a * npx;
      int o = s_act[a];
      int xsrc = min((int)(((int64_t)(x0 + xl) * W) / OW), W - 1);
      if (masks == nullptr)
        s_wx[a * pxc + xl] = coverage(lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]);
      else  // full 2-D weight of this pixel; the row factor below is 1
        s_wx[a * pxc + xl] = mask_weight(masks + ((int64_t)b * O + o) * M * M, M, s_iy0[a], s_wy[a], s_wy1[a],
                                         lin01(xsrc, W), bx[o * 4 + 0], bx[o * 4 + 2]);
    }
    for (int i = tid; i < nact * S; i += 256) {
      int a = i / S, d = i - a * S;
      s_vec[a * S + d] = vv[(int64_t)s_act[a] * S + d];
    }
    __syncthreads();
    for (int a = 0; a < nact; ++a) {
      const float wy = masks == nullptr ? s_wy[a] : 1.0f;
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
  float* orow = out + ((int64_t)(b * OH + y) * OW + x0) * out_cs + out_off;
#pragma unroll
  for (int i = 0; i < LAY_EPT; ++i) {
    if (tid + 256 * i < nel) *(float4*)&orow[(int64_t)ex[i] * out_cs + eq[i] * 4] = acc[i];
  }
}

// One block per (object, image): reduce dout over the box's support only.  256 threads for small layouts,
// 1024 for >= 64x64 ones (S/4 lanes per pixel, blockDim/(S/4) pixels in flight, two loads per lane in flight).
__global__ __launch_bounds__(1024) void k_layout_bwd(const float* __restrict__ dout, int out_cs, int out_off,
                                                     const float* __restrict__ boxes,
                                                     const uint8_t* __restrict__ valid,
                                                     const float* __restrict__ masks, int M, int O, int S, int H,
                                                     int W, int OH, int OW, float* __restrict__ dvecs,
                                                     int accumulate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_wy = sm;            // [OH]
  float* s_wx = sm + OH;       // [OW]
  int* s_rng = (int*)(s_wx + OW);  // ylo, yhi, xlo, xhi
  float* s_red = sm + (((OH + OW + 4) + 3) & ~3);  // [npl][S], 16-byte aligned
  const int tid = threadIdx.x, BD = blockDim.x;
  const int o = blockIdx.x, b = blockIdx.y;
  float* dv = dvecs + ((int64_t)b * O + o) * S;
  if (!valid[(int64_t)b * O + o]) {
    if (!accumulate)
      for (int d = tid; d < S; d += BD) dv[d] = 0.f;
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
    if (w != 0.f) {
      atomicMin(&s_rng[0], y);
      atomicMax(&s_rng[1], y);
    }
  }
  for (int x = tid; x < OW; x += BD) {
    int xsrc = min((int)(((int64_t)x * W) / OW), W - 1);
    float w = masks == nullptr ? coverage(lin01(xsrc, W), x0, ww) : coverage_n(lin01(xsrc, W), x0, ww, M);
    s_wx[x] = w;
    if (w != 0.f) {
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
}

extern "C" {

int csg_layout_fwd(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                   int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                   int64_t out_cs, int64_t out_off, void* stream) {
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
  size_t shm = (size_t)(LAY_OB * pxc + LAY_OB * S + LAY_OB) * 4 + (LAY_OB + 4) * 4 + 2 * LAY_OB * 4;
  ProfScope p(K_LAYOUT_FWD, (double)B * OH * OW * S * 4, s);  // algorithmic bytes: the output, once
  dim3 grid((unsigned)cdiv(OW, pxc), (unsigned)OH, (unsigned)B);
  hipLaunchKernelGGL(k_layout_fwd, grid, dim3(256), shm, s, vecs, boxes, valid, masks, (int)M, (int)O, (int)S, (int)H,
                     (int)W, (int)OH, (int)OW, pxc, out, (int)out_cs, (int)out_off);
  return check_launch("csg_layout_fwd");
}

int csg_layout_bwd(const float* dout, int64_t out_cs, int64_t out_off, const float* boxes, const uint8_t* valid,
                   const float* masks, int64_t M, int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH,
                   int64_t OW, float* dvecs, int accumulate, void* stream) {
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
  size_t shm = (size_t)(((OH + OW + 4) + 3) & ~3) * 4 + (size_t)npl * S * 4;
  ProfScope p(K_LAYOUT_BWD, (double)B * OH * OW * S * 4, s);
  hipLaunchKernelGGL(k_layout_bwd, dim3((unsigned)O, (unsigned)B), dim3((unsigned)bd), shm, s, dout, (int)out_cs, (int)out_off,
                     boxes, valid, masks, (int)M, (int)O, (int)S, (int)H, (int)W, (int)OH, (int)OW, dvecs, accumulate);
  return check_launch("csg_layout_bwd");
}

}  // extern "C"
