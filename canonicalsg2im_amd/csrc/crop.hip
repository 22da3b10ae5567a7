// Differentiable bilinear object crops for the object discriminator
// (reference: sg2im/bilinear.py:44-94 — crop_bbox_batch_cudnn + crop_bbox).
//
// The reference expands the image once per object ((sum O, C, H, W) tensor) and calls
// F.grid_sample; here each output pixel of each crop gathers its 4 source pixels directly from the
// object's own image (NHWC), so the traffic is the crops themselves.  Backward is a GATHER as well
// (crops of different objects overlap in the image): every image pixel sums the taps that touch it over
// the crops of its image in crop order, crop rows, then crop columns — no atomics, the same bits every run.
#include "csg_common.h"

using namespace csg;

// torch.linspace(0,1,n)[i] and torch.linspace(1,0,n)[i] as ATen evaluates them
__device__ __forceinline__ float lin_up(int i, int n) {
  if (n <= 1) return 0.f;
  float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? (float)i * step : 1.0f - (float)(n - 1 - i) * step;
}
__device__ __forceinline__ float lin_down(int i, int n) {
  if (n <= 1) return 1.f;
  float step = -1.0f / (float)(n - 1);
  return (i < n / 2) ? 1.0f + (float)i * step : 0.0f - (float)(n - 1 - i) * step;
}

struct CropTaps {
  int ix0, iy0;
  float w00, w01, w10, w11;  // [y][x]
};

// boxes are [x0,y0,w,h] in [0,1]; crop_bbox maps them to [-1,1] corners and interpolates the
// sampling grid with tensor_linspace (start_w*start + end_w*end), then grid_sample(bilinear, zeros,
// align_corners=False)
__device__ __forceinline__ CropTaps crop_taps(const float* __restrict__ box, int x, int y, int WW, int HH, int W,
                                              int H) {
  const float bx0 = 2.0f * box[0] - 1.0f, by0 = 2.0f * box[1] - 1.0f;
  const float bx1 = 2.0f * (box[0] + box[2]) - 1.0f, by1 = 2.0f * (box[1] + box[3]) - 1.0f;
  const float gx = lin_down(x, WW) * bx0 + lin_up(x, WW) * bx1;
  const float gy = lin_down(y, HH) * by0 + lin_up(y, HH) * by1;
  const float fx = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
  const float fy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
  float x0f = fminf(fmaxf(floorf(fx), -2.0f), (float)W), y0f = fminf(fmaxf(floorf(fy), -2.0f), (float)H);
  const float tx = fx - floorf(fx), ty = fy - floorf(fy);
  CropTaps t;
  t.ix0 = (int)x0f;
  t.iy0 = (int)y0f;
  const bool x0ok = t.ix0 >= 0 && t.ix0 < W, x1ok = t.ix0 + 1 >= 0 && t.ix0 + 1 < W;
  const bool y0ok = t.iy0 >= 0 && t.iy0 < H, y1ok = t.iy0 + 1 >= 0 && t.iy0 + 1 < H;
  t.w00 = (x0ok && y0ok) ? (1.f - tx) * (1.f - ty) : 0.f;
  t.w01 = (x1ok && y0ok) ? tx * (1.f - ty) : 0.f;
  t.w10 = (x0ok && y1ok) ? (1.f - tx) * ty : 0.f;
  t.w11 = (x1ok && y1ok) ? tx * ty : 0.f;
  return t;
}

__global__ void k_crop_fwd(const float* __restrict__ img, int H, int W, int cs, int C, const float* __restrict__ boxes,
                           const int64_t* __restrict__ img_idx, int64_t total, int HH, int WW, int out_cs,
                           float* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(e % WW);
    const int64_t t1 = e / WW;
    const int y = (int)(t1 % HH);
    const int64_t n = t1 / HH;
    const CropTaps t = crop_taps(boxes + n * 4, x, y, WW, HH, W, H);
    const float* base = img + (int64_t)img_idx[n] * H * W * cs;
    float* o = out + e * out_cs;
    for (int c = 0; c < out_cs; ++c) {
      float v = 0.f;
      if (c < C) {
        if (t.w00 != 0.f) v += t.w00 * base[((int64_t)t.iy0 * W + t.ix0) * cs + c];
        if (t.w01 != 0.f) v += t.w01 * base[((int64_t)t.iy0 * W + t.ix0 + 1) * cs + c];
        if (t.w10 != 0.f) v += t.w10 * base[((int64_t)(t.iy0 + 1) * W + t.ix0) * cs + c];
        if (t.w11 != 0.f) v += t.w11 * base[((int64_t)(t.iy0 + 1) * W + t.ix0 + 1) * cs + c];
      }
      o[c] = v;
    }
  }
}

// d(image) as a gather.  The sampling grid of a crop is separable (fx depends on the crop column only, fy on the crop
// row only), so the taps of crop n are two tables of HH / WW entries: (first source row / column, fraction).  A block owns
// a 16 x 16 pixel tile of one image:
//   1. the crops of ITS image whose footprint (conservatively, from the box corners) meets the tile are compacted, in crop
//      order, into an LDS list — 256 candidates per pass, one per thread, wave ballots + a prefix over the four waves
//      (walking the crop list with dependent scalar loads cost a memory latency per crop: 300 - 700 us per launch);
//   2. eight listed crops at a time get their tap tables staged in LDS;
//   3. every thread sums, over the crop rows and columns whose taps land on its pixel, weight * dout — crops outermost,
//      rows, then columns: a fixed order.  The rows / columns worth testing come from the linear form of the grid
//      (f(i) ~ f0 + sl i, widened by one entry each side); the test itself reads the exact staged tables.
// (The scatter form this replaces issued ~3 M float atomics per step; their arrival order decided the last bits of d(image).)
#define CROP_MAXHW 64
#define CROP_K 8
__global__ __launch_bounds__(256) void k_crop_bwd(const float* __restrict__ dout, int H, int W, int cs, int C,
                                                   const float* __restrict__ boxes, const int64_t* __restrict__ img_idx,
                                                   int N, int HH, int WW, int out_cs, int tiles_x, int tiles_per_img,
                                                   float* __restrict__ dimg) {
  __shared__ int s_i0[CROP_K][2][CROP_MAXHW];      // [crop of the group][axis: 0 = x, 1 = y][crop column / row]: first source index
  __shared__ float s_t[CROP_K][2][CROP_MAXHW];     // fraction towards the second one
  __shared__ float s_lin[CROP_K][2][2];            // [axis]: f0, sl of the linear form
  __shared__ int s_list[256];
  __shared__ int s_cnt[5];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.x / tiles_per_img, tile = blockIdx.x - b * tiles_per_img;
  const int tx0 = (tile % tiles_x) * 16, ty0 = (tile / tiles_x) * 16;
  const int x = tx0 + (tid & 15), y = ty0 + (tid >> 4);
  const bool inside = x < W && y < H;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < N; c0 += 256) {
    // ---- 1. ordered compaction of the candidates c0 .. c0 + 255
    const int n = c0 + tid;
    bool take = false;
    if (n < N && img_idx[n] == b) {
      const float4 bx = *(const float4*)(boxes + (int64_t)n * 4);
      // conservative footprint from the two ends of each axis (the grid is a monotone interpolation between them; two
      // pixels of slack for its rounding and the second tap)
      const float fx0 = ((2.0f * bx.x - 1.0f + 1.0f) * (float)W - 1.0f) / 2.0f;
      const float fx1 = ((2.0f * (bx.x + bx.z) - 1.0f + 1.0f) * (float)W - 1.0f) / 2.0f;
      const float fy0 = ((2.0f * bx.y - 1.0f + 1.0f) * (float)H - 1.0f) / 2.0f;
      const float fy1 = ((2.0f * (bx.y + bx.w) - 1.0f + 1.0f) * (float)H - 1.0f) / 2.0f;
      take = fminf(fx0, fx1) - 2.0f <= (float)(tx0 + 15) && fmaxf(fx0, fx1) + 2.0f >= (float)tx0 &&
             fminf(fy0, fy1) - 2.0f <= (float)(ty0 + 15) && fmaxf(fy0, fy1) + 2.0f >= (float)ty0;
    }
    const unsigned long long m = __ballot(take);
    __syncthreads();                               // the previous pass is done with s_list / s_cnt
    if (lane == 0) s_cnt[wv] = __popcll(m);
    __syncthreads();
    if (take) {
      int pos = __popcll(m & ((1ull << lane) - 1ull));
      for (int w2 = 0; w2 < wv; ++w2) pos += s_cnt[w2];
      s_list[pos] = n;
    }
    const int L = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    // ---- 2. / 3. groups of CROP_K listed crops
    for (int k0 = 0; k0 < L; k0 += CROP_K) {
      __syncthreads();                             // s_list complete / the previous group's tables are no longer read
      const int kn = min(CROP_K, L - k0);
      for (int e = tid; e < kn * (WW + HH); e += 256) {
        const int k = e / (WW + HH), r = e - k * (WW + HH);
        const bool isx = r < WW;
        const int i = isx ? r : r - WW, nn = isx ? WW : HH, Ld = isx ? W : H;
        const float* box = boxes + (int64_t)s_list[k0 + k] * 4;
        const float lo = isx ? box[0] : box[1], sz = isx ? box[2] : box[3];
        const float b0 = 2.0f * lo - 1.0f, b1 = 2.0f * (lo + sz) - 1.0f;
        const float g = lin_down(i, nn) * b0 + lin_up(i, nn) * b1;
        const float f = ((g + 1.0f) * (float)Ld - 1.0f) / 2.0f;
        s_i0[k][isx ? 0 : 1][i] = (int)fminf(fmaxf(floorf(f), -2.0f), (float)Ld);
        s_t[k][isx ? 0 : 1][i] = f - floorf(f);
        if (i == 0) {
          s_lin[k][isx ? 0 : 1][0] = ((b0 + 1.0f) * (float)Ld - 1.0f) / 2.0f;
          s_lin[k][isx ? 0 : 1][1] = nn > 1 ? (b1 - b0) * (float)Ld / (2.0f * (float)(nn - 1)) : 0.0f;
        }
      }
      __syncthreads();
      if (inside) {
        for (int k = 0; k < kn; ++k) {
          // candidate rows / columns: f(i) in [p - 1, p + 1)  ->  i in [(p - 1 - f0) / sl, (p + 1 - f0) / sl), widened;
          // a flat or reversed grid (sl <= 0: degenerate box) scans everything
          int ylo = 0, yhi = HH - 1, xlo = 0, xhi = WW - 1;
          const float fy0 = s_lin[k][1][0], sly = s_lin[k][1][1], fx0 = s_lin[k][0][0], slx = s_lin[k][0][1];
          if (sly > 1e-3f) {
            ylo = max(0, (int)fminf(floorf(((float)(y - 1) - fy0) / sly), 1e6f) - 1);
            yhi = min(HH - 1, (int)fmaxf(ceilf(((float)(y + 1) - fy0) / sly), -1e6f) + 1);
          }
          if (slx > 1e-3f) {
            xlo = max(0, (int)fminf(floorf(((float)(x - 1) - fx0) / slx), 1e6f) - 1);
            xhi = min(WW - 1, (int)fmaxf(ceilf(((float)(x + 1) - fx0) / slx), -1e6f) + 1);
          }
          const float* g = dout + (int64_t)s_list[k0 + k] * HH * WW * out_cs;
          for (int cy = ylo; cy <= yhi; ++cy) {
            const int iy0 = s_i0[k][1][cy];
            if (iy0 != y && iy0 + 1 != y) continue;
            const float ty = s_t[k][1][cy];
            const float wy = iy0 == y ? 1.f - ty : ty;         // (y itself is inside the image: the tap is a valid one)
            for (int cx = xlo; cx <= xhi; ++cx) {
              const int ix0 = s_i0[k][0][cx];
              if (ix0 != x && ix0 + 1 != x) continue;
              const float tx = s_t[k][0][cx];
              // the four weights exactly as crop_taps forms them: (1-tx)(1-ty), tx(1-ty), (1-tx)ty, tx ty
              const float w = (ix0 == x ? 1.f - tx : tx) * wy;
              if (w == 0.f) continue;
              const float* gp = g + ((int64_t)cy * WW + cx) * out_cs;
              for (int c = 0; c < C; ++c) acc[c] += w * gp[c];
            }
          }
        }
      }
    }
  }
  if (inside) {
    float* o = dimg + (((int64_t)b * H + y) * W + x) * cs;
    for (int c = 0; c < C; ++c) o[c] += acc[c];
  }
}

// Gradient w.r.t. the boxes (the sampling grid of crop_bbox is differentiable in x0, y0, w, h: bilinear.py:83-94).
// With fx = ((gx + 1) W - 1) / 2 and gx = lin_down(x) (2 x0 - 1) + lin_up(x) (2 (x0 + w) - 1):
//   d out / d fx = (v01 - v00) (1 - ty) + (v11 - v10) ty   (out-of-image taps count as zeros, as in ATen's
//   grid_sampler backward), d fx / d x0 = W (lin_down + lin_up), d fx / d w = W lin_up; likewise in y.
// One block per crop; every thread sums its pixels in order, the block combines the 256 partial sums in thread order:
// bit-reproducible.
__global__ __launch_bounds__(256) void k_crop_bwd_boxes(const float* __restrict__ dout, const float* __restrict__ img,
                                                         int H, int W, int cs, int C, const float* __restrict__ boxes,
                                                         const int64_t* __restrict__ img_idx, int HH, int WW, int out_cs,
                                                         float* __restrict__ dboxes) {
  __shared__ float4 s_red[256];
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* box = boxes + (int64_t)n * 4;
  const float* base = img + (int64_t)img_idx[n] * H * W * cs;
  const float bx0 = 2.0f * box[0] - 1.0f, by0 = 2.0f * box[1] - 1.0f;
  const float bx1 = 2.0f * (box[0] + box[2]) - 1.0f, by1 = 2.0f * (box[1] + box[3]) - 1.0f;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);       // d x0, d y0, d w, d h
  for (int e = tid; e < HH * WW; e += 256) {
    const int y = e / WW, x = e - y * WW;
    const float ldx = lin_down(x, WW), lux = lin_up(x, WW), ldy = lin_down(y, HH), luy = lin_up(y, HH);
    const float gx = ldx * bx0 + lux * bx1, gy = ldy * by0 + luy * by1;
    const float fx = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f, fy = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
    const float x0f = fminf(fmaxf(floorf(fx), -2.0f), (float)W), y0f = fminf(fmaxf(floorf(fy), -2.0f), (float)H);
    const float tx = fx - floorf(fx), ty = fy - floorf(fy);
    const int ix0 = (int)x0f, iy0 = (int)y0f;
    const bool x0ok = ix0 >= 0 && ix0 < W, x1ok = ix0 + 1 >= 0 && ix0 + 1 < W;
    const bool y0ok = iy0 >= 0 && iy0 < H, y1ok = iy0 + 1 >= 0 && iy0 + 1 < H;
    const float* g = dout + ((int64_t)n * HH * WW + e) * out_cs;
    float dfx = 0.f, dfy = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v00 = (x0ok && y0ok) ? base[((int64_t)iy0 * W + ix0) * cs + c] : 0.f;
      const float v01 = (x1ok && y0ok) ? base[((int64_t)iy0 * W + ix0 + 1) * cs + c] : 0.f;
      const float v10 = (x0ok && y1ok) ? base[((int64_t)(iy0 + 1) * W + ix0) * cs + c] : 0.f;
      const float v11 = (x1ok && y1ok) ? base[((int64_t)(iy0 + 1) * W + ix0 + 1) * cs + c] : 0.f;
      const float go = g[c];
      dfx += go * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
      dfy += go * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
    }
    const float ax = dfx * (float)W, ay = dfy * (float)H;       // d fx / d gx = W / 2, d gx / d x0 = 2 (ld + lu), ...
    acc.x += ax * (ldx + lux);
    acc.y += ay * (ldy + luy);
    acc.z += ax * lux;
    acc.w += ay * luy;
  }
  s_red[tid] = acc;
  __syncthreads();
  if (tid == 0) {
    float4 t = s_red[0];
    for (int i = 1; i < 256; ++i) { t.x += s_red[i].x; t.y += s_red[i].y; t.z += s_red[i].z; t.w += s_red[i].w; }
    *(float4*)(dboxes + (int64_t)n * 4) = t;
  }
}

extern "C" {

int csg_crop_bwd_boxes(const float* dout, const float* img, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C,
                       const float* boxes, const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, int64_t out_cs,
                       float* dboxes, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && img_cs >= C && out_cs >= C && HH > 0 && WW > 0 && N >= 0,
              CSG_E_BADSHAPE, "csg_crop_bwd_boxes: bad shape");
  CSG_REQUIRE(((uintptr_t)dboxes % 16) == 0, CSG_E_UNSUPPORTED, "csg_crop_bwd_boxes: dboxes must be 16-byte aligned");
  if (N == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_CROP_BWD, (double)N * HH * WW * (C * 4 + out_cs) * 4, s);
  CSG_LAUNCH(k_crop_bwd_boxes, dim3((unsigned)N), dim3(256), 0, s, dout, img, (int)H, (int)W, (int)img_cs, (int)C, boxes,
             img_idx, (int)HH, (int)WW, (int)out_cs, dboxes);
  return check_launch("csg_crop_bwd_boxes");
}

int csg_crop_fwd(const float* img, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C, const float* boxes,
                 const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, float* out, int64_t out_cs, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && img_cs >= C && out_cs >= C && HH > 0 && WW > 0 && N >= 0,
              CSG_E_BADSHAPE, "csg_crop_fwd: bad shape");
  if (N == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = N * HH * WW;
  ProfScope p(K_CROP_FWD, (double)total * (C * 4 + out_cs) * 4, s);
  int64_t g = cdiv(total, 256);
  if (g > 4096) g = 4096;
  CSG_LAUNCH(k_crop_fwd, dim3((unsigned)g), dim3(256), 0, s, img, (int)H, (int)W, (int)img_cs, (int)C, boxes,
                     img_idx, total, (int)HH, (int)WW, (int)out_cs, out);
  return check_launch("csg_crop_fwd");
}

int csg_crop_bwd(const float* dout, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C, const float* boxes,
                 const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, int64_t out_cs, float* dimg, void* stream) {
  CSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && img_cs >= C && out_cs >= C && HH > 0 && WW > 0 && N >= 0,
              CSG_E_BADSHAPE, "csg_crop_bwd: bad shape");
  if (N == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  CSG_REQUIRE(C <= 4 && HH <= CROP_MAXHW && WW <= CROP_MAXHW, CSG_E_UNSUPPORTED,
              "csg_crop_bwd: at most 4 channels and %d x %d crops", CROP_MAXHW, CROP_MAXHW);
  CSG_REQUIRE(((uintptr_t)boxes % 16) == 0, CSG_E_UNSUPPORTED, "csg_crop_bwd: boxes must be 16-byte aligned");
  const int64_t total = N * HH * WW;
  ProfScope p(K_CROP_BWD, (double)total * (C * 4 + out_cs) * 4, s);
  const int tiles_x = (int)cdiv(W, 16), tiles_per_img = tiles_x * (int)cdiv(H, 16);
  CSG_LAUNCH(k_crop_bwd, dim3((unsigned)(B * tiles_per_img)), dim3(256), 0, s, dout, (int)H, (int)W, (int)img_cs, (int)C, boxes,
             img_idx, (int)N, (int)HH, (int)WW, (int)out_cs, tiles_x, tiles_per_img, dimg);
  return check_launch("csg_crop_bwd");
}

}  // extern "C"
