// Scene-graph encoder kernels (K1, K2, K4, K5): embedding lookup, real-object mask, per-image
// CSR of triplets, gather-concat and the confidence-weighted segment average of
// GraphTripleConv (reference: sg2im/graph.py:44-113, sg2im/attribute_embed.py:31-48).
//
// All of these are HBM/latency-bound index kernels: wave64, one row segment per wave, no
// atomics on the forward path, fixed summation order (subject entries in t order, then object
// entries in t order) so results are reproducible run to run.
#include "csg_common.h"

using namespace csg;

// ------------------------------------------------------------------------------------ K1
__global__ void k_embed_fwd(const int64_t* __restrict__ idx, int64_t rows, int64_t idx_stride,
                            const float* __restrict__ table, int64_t num_emb, int64_t dim, float* __restrict__ out,
                            int64_t out_stride, int64_t out_off) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * dim) return;
  int64_t r = e / dim, d = e - r * dim;
  int64_t i = idx[r * idx_stride];
  float v = (i >= 0 && i < num_emb) ? table[i * dim + d] : __builtin_nanf("");
  out[r * out_stride + out_off + d] = v;
}

__global__ void k_embed_bwd(const int64_t* __restrict__ idx, int64_t rows, int64_t idx_stride,
                            const float* __restrict__ dout, int64_t out_stride, int64_t out_off, int64_t num_emb,
                            int64_t dim, float* __restrict__ dtable) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * dim) return;
  int64_t r = e / dim, d = e - r * dim;
  int64_t i = idx[r * idx_stride];
  if (i >= 0 && i < num_emb) atomicAdd(&dtable[i * dim + d], dout[r * out_stride + out_off + d]);
}

__global__ void k_obj_mask(const int64_t* __restrict__ objs, int64_t n, int64_t A, int64_t image_id,
                           uint8_t* __restrict__ mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t v = objs[i * A];
  mask[i] = (v != 0 && v != image_id) ? 1 : 0;
}

// ------------------------------------------------------------------------------------ CSR
// One block per image.  Thread j owns objects j, j+256, ... (O <= 1024) and scans the triplet
// list three times (count, fill subjects, fill objects); the list is staged through LDS in chunks.
#define CSR_CHUNK 2048
#define CSR_MAXJ 4
__global__ __launch_bounds__(256) void k_csr_build(const int64_t* __restrict__ triplets, int T, int O,
                                                    int32_t* __restrict__ row_ptr, int32_t* __restrict__ col) {
  __shared__ int s_s[CSR_CHUNK];
  __shared__ int s_o[CSR_CHUNK];
  __shared__ int s_cnt[1024 + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* tr = triplets + (int64_t)b * T * 3;
  int32_t* rp = row_ptr + (int64_t)b * (O + 1);
  int32_t* cl = col + (int64_t)b * 2 * T;

  int cnt[CSR_MAXJ];
#pragma unroll
  for (int j = 0; j < CSR_MAXJ; ++j) cnt[j] = 0;
  for (int t0 = 0; t0 < T; t0 += CSR_CHUNK) {
    int n = min(CSR_CHUNK, T - t0);
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
      s_s[i] = (int)tr[(int64_t)(t0 + i) * 3 + 0];
      s_o[i] = (int)tr[(int64_t)(t0 + i) * 3 + 2];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CSR_MAXJ; ++j) {
      int obj = tid + 256 * j;
      if (obj < O) {
        int c = 0;
        for (int i = 0; i < n; ++i) c += (s_s[i] == obj) + (s_o[i] == obj);
        cnt[j] += c;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CSR_MAXJ; ++j) {
    int obj = tid + 256 * j;
    if (obj < O) s_cnt[obj] = cnt[j];
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < O; ++i) {
      int c = s_cnt[i];
      s_cnt[i] = run;
      run += c;
    }
    s_cnt[O] = run;
  }
  __syncthreads();
  for (int i = tid; i <= O; i += 256) rp[i] = s_cnt[i];
  int pos[CSR_MAXJ];
#pragma unroll
  for (int j = 0; j < CSR_MAXJ; ++j) {
    int obj = tid + 256 * j;
    pos[j] = obj < O ? s_cnt[obj] : 0;
  }
  for (int role = 0; role < 2; ++role) {
    for (int t0 = 0; t0 < T; t0 += CSR_CHUNK) {
      int n = min(CSR_CHUNK, T - t0);
      __syncthreads();
      for (int i = tid; i < n; i += 256) s_s[i] = (int)tr[(int64_t)(t0 + i) * 3 + 2 * role];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CSR_MAXJ; ++j) {
        int obj = tid + 256 * j;
        if (obj < O) {
          int p = pos[j];
          for (int i = 0; i < n; ++i)
            if (s_s[i] == obj) cl[p++] = 2 * (t0 + i) + role;
          pos[j] = p;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------ K2
__global__ void k_gather_concat_fwd(const float* __restrict__ obj, const float* __restrict__ pred,
                                    const int64_t* __restrict__ triplets, int64_t BT, int O, int T, int Din, int Dp,
                                    float* __restrict__ out) {
  const int Dc = 2 * Din + Dp;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BT * Dc) return;
  int64_t bt = e / Dc;
  int j = (int)(e - bt * Dc);
  int64_t b = bt / T;
  float v;
  if (j < Din) {
    int64_t s = triplets[bt * 3 + 0];
    v = obj[(b * O + s) * Din + j];
  } else if (j < Din + Dp) {
    v = pred[bt * Dp + (j - Din)];
  } else {
    int64_t o = triplets[bt * 3 + 2];
    v = obj[(b * O + o) * Din + (j - Din - Dp)];
  }
  out[e] = v;
}

// dobj[b,i,:] = sum over the CSR row of the matching slice of dcat; dpred = middle slice.
__global__ __launch_bounds__(128) void k_gather_concat_bwd_obj(const float* __restrict__ dcat,
                                                               const int32_t* __restrict__ row_ptr,
                                                               const int32_t* __restrict__ col, int O, int T, int Din,
                                                               int Dp, float* __restrict__ dobj) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int Dc = 2 * Din + Dp;
  const int32_t* rp = row_ptr + (int64_t)b * (O + 1);
  const int32_t* cl = col + (int64_t)b * 2 * T;
  const int beg = rp[i], end = rp[i + 1];
  const float* base = dcat + (int64_t)b * T * Dc;
  for (int d = threadIdx.x; d < Din; d += blockDim.x) {
    float acc = 0.f;
    for (int e = beg; e < end; ++e) {
      int c = cl[e];
      int t = c >> 1;
      int off = (c & 1) ? (Din + Dp) : 0;
      acc += base[(int64_t)t * Dc + off + d];
    }
    dobj[((int64_t)b * O + i) * Din + d] = acc;
  }
}

__global__ void k_slice_copy(const float* __restrict__ src, int64_t rows, int src_stride, int src_off, int width,
                             float* __restrict__ dst) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * width) return;
  int64_t r = e / width;
  int j = (int)(e - r * width);
  dst[e] = src[r * src_stride + src_off + j];
}

// ------------------------------------------------------------------------------------ K4+K5
__global__ __launch_bounds__(128) void k_segment_avg_fwd(const float* __restrict__ h, const float* __restrict__ conf,
                                                          const uint8_t* __restrict__ valid,
                                                          const int32_t* __restrict__ row_ptr,
                                                          const int32_t* __restrict__ col, int O, int T, int H, int Dp,
                                                          float* __restrict__ pooled, float* __restrict__ cnt_out) {
  const int i = blockIdx.x, b = blockIdx.y;
  const int Dh = 2 * H + Dp;
  const int32_t* rp = row_ptr + (int64_t)b * (O + 1);
  const int32_t* cl = col + (int64_t)b * 2 * T;
  const int beg = rp[i], end = rp[i + 1];
  const float* hb = h + (int64_t)b * T * Dh;
  const float* cb = conf + (int64_t)b * T;
  const uint8_t* vb = valid + (int64_t)b * T;
  float cnt = 0.f;
  for (int e = beg; e < end; ++e) {
    int t = cl[e] >> 1;
    if (vb[t]) cnt += cb[t];
  }
  const float inv = cnt > 0.f ? 1.0f / cnt : 0.f;
  for (int d = threadIdx.x; d < H; d += blockDim.x) {
    float acc = 0.f;
    for (int e = beg; e < end; ++e) {
      int c = cl[e];
      int t = c >> 1;
      if (vb[t]) {
        int off = (c & 1) ? (H + Dp) : 0;
        acc += hb[(int64_t)t * Dh + off + d] * cb[t];
      }
    }
    // sg2im/graph.py:105-106: divide only where count > 0 (a true division, like the reference)
    pooled[((int64_t)b * O + i) * H + d] = cnt > 0.f ? acc / cnt : acc;
  }
  (void)inv;
  if (threadIdx.x == 0) cnt_out[(int64_t)b * O + i] = cnt;
}

__global__ void k_scale_slice(const float* __restrict__ h, const float* __restrict__ conf, int64_t BT, int Dh,
                              int off, int Dp, float* __restrict__ out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BT * Dp) return;
  int64_t bt = e / Dp;
  int j = (int)(e - bt * Dp);
  out[e] = h[bt * Dh + off + j] * conf[bt];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dcnt[b,i] = -(dpooled . pooled)/cnt where cnt > 0  (divisor path of the average)
__global__ __launch_bounds__(64) void k_segavg_dcnt(const float* __restrict__ dpooled,
                                                     const float* __restrict__ pooled, const float* __restrict__ cnt,
                                                     int H, float* __restrict__ dcnt) {
  const int64_t bi = blockIdx.x;
  float acc = 0.f;
  for (int d = threadIdx.x; d < H; d += 64) acc += dpooled[bi * H + d] * pooled[bi * H + d];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    float c = cnt[bi];
    dcnt[bi] = c > 0.f ? -acc / c : 0.f;
  }
}

// one wave per triplet
__global__ __launch_bounds__(64) void k_segment_avg_bwd(const float* __restrict__ dpooled,
                                                         const float* __restrict__ dnew_p, const float* __restrict__ h,
                                                         const float* __restrict__ conf,
                                                         const uint8_t* __restrict__ valid,
                                                         const int64_t* __restrict__ triplets,
                                                         const float* __restrict__ cnt, const float* __restrict__ dcnt,
                                                         int O, int T, int H, int Dp, float* __restrict__ dh,
                                                         float* __restrict__ dconf) {
  const int64_t bt = blockIdx.x;
  const int64_t b = bt / T;
  const int Dh = 2 * H + Dp;
  const float c = conf[bt];
  const bool v = valid[bt] != 0;
  const int64_t s = triplets[bt * 3 + 0], o = triplets[bt * 3 + 2];
  const bool s_ok = v && s >= 0 && s < O, o_ok = v && o >= 0 && o < O;
  const float cs = s_ok ? cnt[b * O + s] : 0.f, co = o_ok ? cnt[b * O + o] : 0.f;
  const float sc_s = s_ok ? (cs > 0.f ? 1.0f / cs : 1.0f) : 0.f;
  const float sc_o = o_ok ? (co > 0.f ? 1.0f / co : 1.0f) : 0.f;
  const float* hp = h + bt * Dh;
  float* dhp = dh + bt * Dh;
  const float* dps = dpooled + (b * O + (s_ok ? s : 0)) * H;
  const float* dpo = dpooled + (b * O + (o_ok ? o : 0)) * H;
  float acc = 0.f;
  for (int d = threadIdx.x; d < H; d += 64) {
    float gs = dps[d] * sc_s;
    float go = dpo[d] * sc_o;
    dhp[d] = gs * c;
    dhp[H + Dp + d] = go * c;
    acc += gs * hp[d] + go * hp[H + Dp + d];
  }
  for (int d = threadIdx.x; d < Dp; d += 64) {
    float gp = dnew_p ? dnew_p[bt * Dp + d] : 0.f;
    dhp[H + d] = gp * c;
    acc += gp * hp[H + d];
  }
  acc = wave_sum(acc);
  if (threadIdx.x == 0) {
    float extra = 0.f;
    if (s_ok) extra += dcnt[b * O + s];
    if (o_ok) extra += dcnt[b * O + o];
    dconf[bt] = acc + extra;
  }
}

// ------------------------------------------------------------------------------------ C ABI
extern "C" {

int csg_embed_fwd(const int64_t* idx, int64_t rows, int64_t idx_stride, const float* table, int64_t num_emb,
                  int64_t dim, float* out, int64_t out_stride, int64_t out_off, void* stream) {
  CSG_REQUIRE(rows >= 0 && dim > 0 && num_emb > 0, CSG_E_BADSHAPE, "csg_embed_fwd: bad shape");
  if (rows == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_EMBED_FWD, (double)rows * dim * 8, s);
  int64_t n = rows * dim;
  hipLaunchKernelGGL(k_embed_fwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, idx, rows, idx_stride, table, num_emb,
                     dim, out, out_stride, out_off);
  return check_launch("csg_embed_fwd");
}

int csg_embed_bwd(const int64_t* idx, int64_t rows, int64_t idx_stride, const float* dout, int64_t out_stride,
                  int64_t out_off, int64_t num_emb, int64_t dim, float* dtable, void* stream) {
  CSG_REQUIRE(rows >= 0 && dim > 0 && num_emb > 0, CSG_E_BADSHAPE, "csg_embed_bwd: bad shape");
  if (rows == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_EMBED_BWD, (double)rows * dim * 8, s);
  int64_t n = rows * dim;
  hipLaunchKernelGGL(k_embed_bwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, idx, rows, idx_stride, dout,
                     out_stride, out_off, num_emb, dim, dtable);
  return check_launch("csg_embed_bwd");
}

int csg_real_object_mask(const int64_t* objs, int64_t B, int64_t O, int64_t A, int64_t image_id, uint8_t* mask,
                         void* stream) {
  CSG_REQUIRE(B >= 0 && O >= 0 && A > 0, CSG_E_BADSHAPE, "csg_real_object_mask: bad shape");
  if (B * O == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_OBJ_MASK, (double)B * O * 9, s);
  hipLaunchKernelGGL(k_obj_mask, dim3((unsigned)cdiv(B * O, 256)), dim3(256), 0, s, objs, B * O, A, image_id, mask);
  return check_launch("csg_real_object_mask");
}

int csg_graph_csr_build(const int64_t* triplets, int64_t B, int64_t T, int64_t O, int32_t* row_ptr, int32_t* col,
                        void* stream) {
  CSG_REQUIRE(B > 0 && T >= 0 && O > 0, CSG_E_BADSHAPE, "csg_graph_csr_build: bad shape B=%ld T=%ld O=%ld", (long)B,
              (long)T, (long)O);
  CSG_REQUIRE(O <= 256 * CSR_MAXJ, CSG_E_UNSUPPORTED, "csg_graph_csr_build: O=%ld > %d objects per image", (long)O,
              256 * CSR_MAXJ);
  CSG_REQUIRE(T < (1 << 29), CSG_E_UNSUPPORTED, "csg_graph_csr_build: T too large");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_CSR_BUILD, (double)B * T * (24 + 8), s);
  hipLaunchKernelGGL(k_csr_build, dim3((unsigned)B), dim3(256), 0, s, triplets, (int)T, (int)O, row_ptr, col);
  return check_launch("csg_graph_csr_build");
}

int csg_gather_concat_fwd(const float* obj, const float* pred, const int64_t* triplets, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* out, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && Din > 0 && Dp > 0, CSG_E_BADSHAPE, "csg_gather_concat_fwd: bad shape");
  if (T == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t n = B * T * (2 * Din + Dp);
  ProfScope p(K_GATHER_FWD, (double)n * 8, s);
  hipLaunchKernelGGL(k_gather_concat_fwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, obj, pred, triplets, B * T,
                     (int)O, (int)T, (int)Din, (int)Dp, out);
  return check_launch("csg_gather_concat_fwd");
}

int csg_gather_concat_bwd(const float* dcat, const int32_t* row_ptr, const int32_t* col, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* dobj, float* dpred, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && Din > 0 && Dp > 0, CSG_E_BADSHAPE, "csg_gather_concat_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_GATHER_BWD, (double)B * T * (2 * Din + Dp) * 8, s);
  if (dobj) hipLaunchKernelGGL(k_gather_concat_bwd_obj, dim3((unsigned)O, (unsigned)B), dim3(128), 0, s, dcat, row_ptr,
                               col, (int)O, (int)T, (int)Din, (int)Dp, dobj);
  if (dpred && T > 0) {
    int64_t n = B * T * Dp;
    hipLaunchKernelGGL(k_slice_copy, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, dcat, B * T,
                       (int)(2 * Din + Dp), (int)Din, (int)Dp, dpred);
  }
  return check_launch("csg_gather_concat_bwd");
}

int csg_segment_avg_fwd(const float* h, const float* conf, const uint8_t* valid, const int32_t* row_ptr,
                        const int32_t* col, int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, float* pooled,
                        float* cnt, float* new_p, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && H > 0 && Dp >= 0, CSG_E_BADSHAPE, "csg_segment_avg_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: messages 2*T*H*4 + indices + confidence, pooled O*H*4 written
  ProfScope p(K_SEGAVG_FWD, (double)B * (T * (2.0 * H * 4 + 16 + 4) + O * H * 4.0), s);
  hipLaunchKernelGGL(k_segment_avg_fwd, dim3((unsigned)O, (unsigned)B), dim3(128), 0, s, h, conf, valid, row_ptr, col,
                     (int)O, (int)T, (int)H, (int)Dp, pooled, cnt);
  if (new_p && Dp > 0 && T > 0) {
    int64_t n = B * T * Dp;
    hipLaunchKernelGGL(k_scale_slice, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, h, conf, B * T,
                       (int)(2 * H + Dp), (int)H, (int)Dp, new_p);
  }
  return check_launch("csg_segment_avg_fwd");
}

int csg_segment_avg_bwd(const float* dpooled, const float* dnew_p, const float* h, const float* conf,
                        const uint8_t* valid, const int64_t* triplets, const float* pooled, const float* cnt,
                        int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, float* dh, float* dconf,
                        float* dcnt_scratch, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && H > 0 && Dp >= 0, CSG_E_BADSHAPE, "csg_segment_avg_bwd: bad shape");
  if (T == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_SEGAVG_BWD, (double)B * T * (2.0 * H + Dp) * 12, s);
  hipLaunchKernelGGL(k_segavg_dcnt, dim3((unsigned)(B * O)), dim3(64), 0, s, dpooled, pooled, cnt, (int)H,
                     dcnt_scratch);
  hipLaunchKernelGGL(k_segment_avg_bwd, dim3((unsigned)(B * T)), dim3(64), 0, s, dpooled, dnew_p, h, conf, valid,
                     triplets, cnt, dcnt_scratch, (int)O, (int)T, (int)H, (int)Dp, dh, dconf);
  return check_launch("csg_segment_avg_bwd");
}

}  // extern "C"
