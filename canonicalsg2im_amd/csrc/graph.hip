// Scene-graph encoder kernels (K1, K2, K4, K5): embedding lookup, real-object mask, per-image
// CSR of triplets, gather-concat and the confidence-weighted segment average of
// GraphTripleConv (reference: sg2im/graph.py:44-113, sg2im/attribute_embed.py:31-48).
//
// All of these are HBM/latency-bound index kernels: wave64, one row segment per wave, no
// atomics (forward or backward), fixed summation order (subject entries in t order, then object
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

// d(table)[i] = sum over the rows r with idx[r] == i of dout[r], in row order: no atomics, the same bits every run.
// A block owns a chunk of consecutive rows — their indices AND their dout rows staged in LDS with independent, coalesced
// loads (a conditional global load per matching row serialises one memory latency per match: 154 us per launch on dense
// graphs) — x 256 consecutive table entries e = i * dim + d; a thread walks the chunk in order and adds the rows that
// name its table row (a select, so the LDS reads pipeline).  One chunk: the sums go straight onto dtable; more: per-chunk
// partials in the caller's workspace + k_embed_bwd_sum adding them in chunk order.  (Dense graphs send 1e5 predicate
// rows to an 8-row table: 370 chunks of 256 rows.)
#define EMB_TILE_FLOATS 8192
__host__ __device__ inline int emb_chunk_rows(int64_t dim) {
  int64_t r = EMB_TILE_FLOATS / dim;
  return (int)(r > 1024 ? 1024 : (r < 32 ? 32 : r));
}
__global__ __launch_bounds__(256) void k_embed_bwd(const int64_t* __restrict__ idx, int64_t rows, int64_t idx_stride,
                                                    const float* __restrict__ dout, int64_t out_stride, int64_t out_off,
                                                    int num_emb, int dim, int chunk, float* __restrict__ dst,
                                                    int64_t dst_chunk_stride, int accumulate) {
  extern __shared__ float s_d[];                   // [chunk][dim] dout rows, then [chunk] indices
  int* s_idx = (int*)(s_d + (size_t)chunk * dim);
  const int64_t r0 = (int64_t)blockIdx.x * chunk;
  const int n = (int)min((int64_t)chunk, rows - r0);
  for (int j = threadIdx.x; j < n; j += 256) {
    const int64_t v = idx[(r0 + j) * idx_stride];
    s_idx[j] = (v >= 0 && v < num_emb) ? (int)v : -1;
  }
  for (int t = threadIdx.x; t < n * dim; t += 256) {
    const int j = t / dim, d = t - j * dim;
    s_d[t] = dout[(r0 + j) * out_stride + out_off + d];
  }
  __syncthreads();
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= num_emb * dim) return;
  const int i = e / dim, d = e - i * dim;
  float acc = 0.f;
  for (int j = 0; j < n; ++j) acc += s_idx[j] == i ? s_d[j * dim + d] : 0.f;
  float* o = dst + (int64_t)blockIdx.x * dst_chunk_stride + e;
  *o = accumulate ? *o + acc : acc;
}

__global__ void k_embed_bwd_sum(const float* __restrict__ part, int nchunks, int n, float* __restrict__ dtable) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float v = dtable[e];
  for (int c = 0; c < nchunks; ++c) v += part[(int64_t)c * n + e];
  dtable[e] = v;
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

// Counting-sort variant for dense graphs (O <= 254, a few 10^4 triplets per image): the (s, o) columns are
// staged in LDS as bytes, 64 chunk owners count their contiguous range of triplets into a per-(role, object,
// chunk) table, a prefix over chunks turns the table into write offsets, and the owners walk their range a
// second time to emit the entries — O(T) work per image instead of O(T*O), same output order (inside a row:
// subject entries in triplet order, then object entries in triplet order).
#define CSR_NCH 64
__global__ __launch_bounds__(256) void k_csr_build_sorted(const int64_t* __restrict__ triplets, int T, int O,
                                                           int32_t* __restrict__ row_ptr, int32_t* __restrict__ col) {
  extern __shared__ unsigned char lds[];
  uint16_t* tab = (uint16_t*)lds;                               // [2][O][CSR_NCH]
  int* tot = (int*)(lds + (size_t)2 * O * CSR_NCH * 2);        // [2*O]
  int* rows = tot + 2 * O;                                      // [O+1]
  unsigned char* so = (unsigned char*)(rows + O + 1);           // [T][2]: s, o (255 = not an object of this image)
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* tr = triplets + (int64_t)b * T * 3;
  int32_t* rp = row_ptr + (int64_t)b * (O + 1);
  int32_t* cl = col + (int64_t)b * 2 * T;
  for (int i = tid; i < 2 * O * CSR_NCH; i += 256) tab[i] = 0;
  for (int t = tid; t < T; t += 256) {
    const int64_t sv = tr[(int64_t)t * 3], ov = tr[(int64_t)t * 3 + 2];
    so[2 * t] = (sv >= 0 && sv < O) ? (unsigned char)sv : 255;
    so[2 * t + 1] = (ov >= 0 && ov < O) ? (unsigned char)ov : 255;
  }
  __syncthreads();
  const int L = (T + CSR_NCH - 1) / CSR_NCH;
  const int t0 = min(T, tid * L), t1 = min(T, t0 + L);
  if (tid < CSR_NCH) {
    for (int t = t0; t < t1; ++t) {
      const int sv = so[2 * t], ov = so[2 * t + 1];
      if (sv != 255) tab[sv * CSR_NCH + tid]++;
      if (ov != 255) tab[(O + ov) * CSR_NCH + tid]++;
    }
  }
  __syncthreads();
  for (int task = tid; task < 2 * O; task += 256) {              // exclusive prefix over the chunks
    int run = 0;
    for (int c = 0; c < CSR_NCH; ++c) {
      const int v = tab[task * CSR_NCH + c];
      tab[task * CSR_NCH + c] = (uint16_t)run;
      run += v;
    }
    tot[task] = run;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int i = 0; i < O; ++i) {
      rows[i] = run;
      run += tot[i] + tot[O + i];
    }
    rows[O] = run;
  }
  __syncthreads();
  for (int i = tid; i <= O; i += 256) rp[i] = rows[i];
  if (tid < CSR_NCH) {
    for (int t = t0; t < t1; ++t) {
      const int sv = so[2 * t], ov = so[2 * t + 1];
      if (sv != 255) cl[rows[sv] + tab[sv * CSR_NCH + tid]++] = 2 * t;
      if (ov != 255) cl[rows[ov] + tot[ov] + tab[(O + ov) * CSR_NCH + tid]++] = 2 * t + 1;
    }
  }
}

static inline size_t csr_sorted_lds(int64_t T, int64_t O) {
  return (size_t)2 * O * CSR_NCH * 2 + (size_t)(2 * O + O + 1) * 4 + (size_t)2 * T;
}

// ------------------------------------------------------------------------------------ K2
// one float4 per thread (Din, Dp are multiples of 4: host-checked), one row = Dc / 4 quads
__global__ __launch_bounds__(256) void k_gather_concat_fwd(const float* __restrict__ obj, const float* __restrict__ pred,
                                                            const int64_t* __restrict__ triplets, int64_t BT, int O, int T,
                                                            int Din, int Dp, float* __restrict__ out) {
  const int Qc = (2 * Din + Dp) >> 2, Qi = Din >> 2, Qp = Dp >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BT * Qc) return;
  const int64_t bt = e / Qc;
  const int j = (int)(e - bt * Qc);
  const int64_t b = bt / T;
  float4 v;
  if (j < Qi) {
    const int64_t s = triplets[bt * 3 + 0];
    v = *(const float4*)(obj + (b * O + s) * Din + j * 4);
  } else if (j < Qi + Qp) {
    v = *(const float4*)(pred + bt * Dp + (j - Qi) * 4);
  } else {
    const int64_t o = triplets[bt * 3 + 2];
    v = *(const float4*)(obj + (b * O + o) * Din + (j - Qi - Qp) * 4);
  }
  *(float4*)(out + e * 4) = v;
}

// ---- CSR row sums (K5 forward, K2 backward) ---------------------------------------------------
// out[b,i,0:D] = sum over the CSR row of object i of  w_e * src[t_e, off(role_e) + 0:D]
//   role 0 (subject) reads the slice at off0, role 1 (object) at off1;
//   WEIGHTED: w_e = conf[t_e] for valid triplets (invalid ones are skipped) and cnt = sum w_e;
//   otherwise w_e = 1.
// The reference's scatter_add loops (graph.py:98-106) give dense graphs rows of several hundred
// edges (CLEVR closure graphs: ~2*(O-1) per object) and hub rows of thousands (the padded triplets of
// a short sample all point at object 0), so rows are cut into 64-edge segments, one workgroup each;
// inside a workgroup 256/LPE edge groups of LPE lanes work on different edges, a lane owns one float4
// of the D columns, 4 edges are in flight per lane.  Group partials are combined through LDS and
// segment partials by k_rowsum_finish, both in a fixed order: results are bit-reproducible run to run.
#define ROWSUM_SEG_DEFAULT 128       // edges per segment in edge-balanced mode (CSG_ROWSUM_SEG: developer knob; 64 / 128 / 256
                                     // measured 112 / 102 / 135 us per forward launch on config C5)
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_csr_rowsum(const float* __restrict__ src, const float* __restrict__ conf,
                                                     const uint8_t* __restrict__ valid,
                                                     const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ col, int O, int T, int D, int stride,
                                                     int off0, int off1, int LPE, int nseg_max, int ROWSUM_SEG,
                                                     float* __restrict__ out, float* __restrict__ cnt_out,
                                                     float* __restrict__ part, float* __restrict__ part_cnt,
                                                     int32_t* __restrict__ seg_info) {
  __shared__ float4 sm[256];
  __shared__ float smc[256];
  __shared__ int s_seg[1024 + 1];
  __shared__ int s_scan[256];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int G = 256 / LPE, lane = tid % LPE, grp = tid / LPE;
  const int32_t* rp = row_ptr + (int64_t)b * (O + 1);
  const int32_t* cl = col + (int64_t)b * 2 * T;
  int i, sp = 0, S = 1, e0, e1;
  int64_t pidx = 0;                       // index of this block's partial
  if (nseg_max == 0) {                    // sparse graphs: one workgroup per row
    i = blockIdx.x;
    e0 = rp[i];
    e1 = rp[i + 1];
  } else {
    // Edge-balanced mode: rows are cut into segments of ROWSUM_SEG edges (an empty row keeps one, so that it
    // is written) and block g takes segment g of the image, whatever row it belongs to — a hub row (the
    // __image__ object, or object 0 of a heavily padded sample) is spread over as many workgroups as it
    // needs.  Every block rebuilds the row -> first-segment prefix (O <= 1024 ints) and bisects it.
    int n4[4], run = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tid * 4 + q;
      n4[q] = r < O ? max(1, (rp[r + 1] - rp[r] + ROWSUM_SEG - 1) / ROWSUM_SEG) : 0;
      run += n4[q];
    }
    s_scan[tid] = run;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int t = tid >= o ? s_scan[tid - o] : 0;
      __syncthreads();
      s_scan[tid] += t;
      __syncthreads();
    }
    int ex = s_scan[tid] - run;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = tid * 4 + q;
      if (r < O) s_seg[r] = ex;
      ex += n4[q];
    }
    if (tid == 255) s_seg[O] = s_scan[255];
    __syncthreads();
    const int g = blockIdx.x;
    if (g >= s_seg[O]) return;            // uniform
    int lo = 0, hi = O;                   // largest i with s_seg[i] <= g
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_seg[mid] <= g) lo = mid; else hi = mid;
    }
    i = lo;
    sp = g - s_seg[i];
    S = s_seg[i + 1] - s_seg[i];
    e0 = rp[i] + sp * ROWSUM_SEG;
    e1 = min(rp[i + 1], e0 + ROWSUM_SEG);
    pidx = (int64_t)b * nseg_max + g;
    if (sp == 0 && tid == 0) {            // tell the finish pass where the row's partials are
      seg_info[((int64_t)b * O + i) * 2 + 0] = g;
      seg_info[((int64_t)b * O + i) * 2 + 1] = S;
    }
  }
  const float* base = src + (int64_t)b * T * stride;
  const float* cb = WEIGHTED ? conf + (int64_t)b * T : nullptr;
  const uint8_t* vb = WEIGHTED ? valid + (int64_t)b * T : nullptr;
  const int64_t orow = (int64_t)b * O + i;
  float* dst = S > 1 ? part + pidx * D : out + orow * D;

  float cnt = 0.f;
  for (int d0 = 0; d0 < D; d0 += LPE * 4) {
    const int d = d0 + lane * 4;
    const bool live = d < D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int e = e0 + grp;
    for (; e + 3 * G < e1; e += 4 * G) {                 // 4 independent 16-byte loads per lane
      int c[4];
      float w[4];
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = cl[e + u * G];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c[u] >> 1;
        w[u] = WEIGHTED ? (vb[t] ? cb[t] : 0.f) : 1.f;
        const bool take = live && (!WEIGHTED || vb[t]);
        v[u] = take ? *(const float4*)(base + (int64_t)t * stride + ((c[u] & 1) ? off1 : off0) + d)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc.x += v[u].x * w[u]; acc.y += v[u].y * w[u]; acc.z += v[u].z * w[u]; acc.w += v[u].w * w[u];
        if (WEIGHTED && d0 == 0) cnt += w[u];
      }
    }
    for (; e < e1; e += G) {
      const int c = cl[e], t = c >> 1;
      const float w = WEIGHTED ? (vb[t] ? cb[t] : 0.f) : 1.f;
      if (live && (!WEIGHTED || vb[t])) {
        const float4 v = *(const float4*)(base + (int64_t)t * stride + ((c & 1) ? off1 : off0) + d);
        acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
      }
      if (WEIGHTED && d0 == 0) cnt += w;
    }
    __syncthreads();
    sm[tid] = acc;
    if (d0 == 0) smc[tid] = cnt;
    __syncthreads();
    if (grp == 0) {
      float4 tot = sm[lane];
      for (int g = 1; g < G; ++g) {                      // fixed order over the edge groups
        const float4 o = sm[g * LPE + lane];
        tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w;
      }
      float ctot = 0.f;
      if (WEIGHTED) {
        for (int g = 0; g < G; ++g) ctot += smc[g * LPE];
        if (d0 == 0) cnt = ctot;                        // lanes of group 0 now hold the chunk's total
        ctot = cnt;
        if (S == 1 && ctot > 0.f) {                     // graph.py:105-106: a true division where count > 0
          tot.x /= ctot; tot.y /= ctot; tot.z /= ctot; tot.w /= ctot;
        }
      }
      if (live) *(float4*)(dst + d) = tot;
    }
  }
  if (WEIGHTED && tid == 0) {
    if (S > 1) part_cnt[pidx] = cnt;
    else cnt_out[orow] = cnt;
  }
}

// rows that were cut into several segments: combine their partials (rows with one segment were written
// directly by k_csr_rowsum).  1024 threads = 1024/LPE groups of LPE lanes; group g adds segments g, g+G, ...
// (a hub row has hundreds), the groups are combined through LDS in a fixed order.
template <bool WEIGHTED>
__global__ __launch_bounds__(1024) void k_rowsum_finish(const float* __restrict__ part,
                                                         const float* __restrict__ part_cnt,
                                                         const int32_t* __restrict__ seg_info, int O, int D, int LPE,
                                                         int nseg_max, float* __restrict__ out,
                                                         float* __restrict__ cnt_out) {
  __shared__ float4 sm[1024];
  const int64_t orow = blockIdx.x;
  const int g0 = seg_info[orow * 2], S = seg_info[orow * 2 + 1];
  if (S <= 1) return;
  const int tid = threadIdx.x, G = 1024 / LPE, lane = tid % LPE, grp = tid / LPE;
  const int64_t p0 = (orow / O) * nseg_max + g0;
  float cnt = 0.f;
  if (WEIGHTED)
    for (int sp = 0; sp < S; ++sp) cnt += part_cnt[p0 + sp];          // S floats, same order in every thread
  for (int d0 = 0; d0 < D; d0 += LPE * 4) {
    const int d = d0 + lane * 4;
    const bool live = d < D;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    int sp = grp;
    for (; sp + G < S; sp += 2 * G) {
      if (live) {
        const float4 u = *(const float4*)(part + (p0 + sp) * D + d);
        const float4 v = *(const float4*)(part + (p0 + sp + G) * D + d);
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
      }
    }
    if (sp < S && live) {
      const float4 u = *(const float4*)(part + (p0 + sp) * D + d);
      a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
    }
    a0.x += a1.x; a0.y += a1.y; a0.z += a1.z; a0.w += a1.w;
    __syncthreads();
    sm[tid] = a0;
    __syncthreads();
    if (grp == 0 && live) {
      float4 tot = sm[lane];
      for (int g = 1; g < G; ++g) {
        const float4 o = sm[g * LPE + lane];
        tot.x += o.x; tot.y += o.y; tot.z += o.z; tot.w += o.w;
      }
      if (WEIGHTED && cnt > 0.f) {
        tot.x /= cnt; tot.y /= cnt; tot.z /= cnt; tot.w /= cnt;
      }
      *(float4*)(out + orow * D + d) = tot;
    }
  }
  if (WEIGHTED && tid == 0) cnt_out[orow] = cnt;
}

// lanes per edge (a power of two <= 256 covering D/4 float4 columns) and row splits for B*O rows of
// average degree 2T/O
static inline int rowsum_lpe(int64_t D) {
  int l = 1;
  while (l * 4 < D && l < 256) l <<= 1;
  return l;
}
// segments per image in edge-balanced mode (0 = sparse graph: one workgroup per row).  Dense mode is chosen
// from the PADDED triplet count, so a batch whose longest sample is dense uses it for every sample.
static inline int rowsum_seg() {
  static const int seg = getenv("CSG_ROWSUM_SEG") ? atoi(getenv("CSG_ROWSUM_SEG")) : ROWSUM_SEG_DEFAULT;
  return seg >= 16 ? seg : ROWSUM_SEG_DEFAULT;
}
static inline int64_t rowsum_nseg(int64_t O, int64_t T) {
  const int64_t deg = O > 0 ? (2 * T + O - 1) / O : 0;
  if (deg <= 48 || O > 1024) return 0;
  return cdiv(2 * T, rowsum_seg()) + O;
}
// workspace: partial rows (+ counts when weighted) and the (first segment, segments) pair of every row
static inline int64_t rowsum_ws_bytes(int64_t B, int64_t O, int64_t T, int64_t D, bool weighted) {
  const int64_t ns = rowsum_nseg(O, T);
  if (ns == 0) return 0;
  return (B * ns * (D + (weighted ? 1 : 0))) * (int64_t)sizeof(float) + B * O * 2 * (int64_t)sizeof(int32_t);
}

// float4 per thread: widths, strides and offsets are multiples of 4 (host-checked)
__global__ __launch_bounds__(256) void k_slice_copy(const float* __restrict__ src, int64_t rows, int src_stride,
                                                     int src_off, int width, float* __restrict__ dst) {
  const int Qw = width >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * Qw) return;
  const int64_t r = e / Qw;
  const int j = (int)(e - r * Qw);
  *(float4*)(dst + e * 4) = *(const float4*)(src + r * src_stride + src_off + j * 4);
}

// ------------------------------------------------------------------------------------ K4+K5
__global__ __launch_bounds__(256) void k_scale_slice(const float* __restrict__ h, const float* __restrict__ conf,
                                                      int64_t BT, int Dh, int off, int Dp, float* __restrict__ out) {
  const int Qp = Dp >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= BT * Qp) return;
  const int64_t bt = e / Qp;
  const int j = (int)(e - bt * Qp);
  const float c = conf[bt];
  float4 v = *(const float4*)(h + bt * Dh + off + j * 4);
  v.x *= c; v.y *= c; v.z *= c; v.w *= c;
  *(float4*)(out + e * 4) = v;
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
                                                         int O, int T, int H, int Dp, int gate_relu,
                                                         float* __restrict__ dh, float* __restrict__ dconf) {
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
  // 16 bytes per lane (H, Dp and the row stride are multiples of 4: host-checked): a quarter of the memory instructions of
  // the dword form this replaces (round 5: 234 -> 148 us per launch on config C5's 94 500 triplets, 44 % -> 69 % of the HBM peak)
  float acc = 0.f;
  const float cg = c;
  auto gate4 = [&](const float4& hv, float4 g) {
    if (gate_relu) {
      if (!(hv.x > 0.f)) g.x = 0.f;
      if (!(hv.y > 0.f)) g.y = 0.f;
      if (!(hv.z > 0.f)) g.z = 0.f;
      if (!(hv.w > 0.f)) g.w = 0.f;
    }
    return g;
  };
  for (int d = threadIdx.x * 4; d < H; d += 256) {
    const float4 a = *(const float4*)(dps + d), bq = *(const float4*)(dpo + d);
    const float4 hs = *(const float4*)(hp + d), ho = *(const float4*)(hp + H + Dp + d);
    const float4 gs = make_float4(a.x * sc_s, a.y * sc_s, a.z * sc_s, a.w * sc_s);
    const float4 go = make_float4(bq.x * sc_o, bq.y * sc_o, bq.z * sc_o, bq.w * sc_o);
    // gate_relu: h is the output of a ReLU and dh goes to that ReLU's producer as the gradient of its PRE-activation
    *(float4*)(dhp + d) = gate4(hs, make_float4(gs.x * cg, gs.y * cg, gs.z * cg, gs.w * cg));
    *(float4*)(dhp + H + Dp + d) = gate4(ho, make_float4(go.x * cg, go.y * cg, go.z * cg, go.w * cg));
    acc += gs.x * hs.x + go.x * ho.x;
    acc += gs.y * hs.y + go.y * ho.y;
    acc += gs.z * hs.z + go.z * ho.z;
    acc += gs.w * hs.w + go.w * ho.w;
  }
  for (int d = threadIdx.x * 4; d < Dp; d += 256) {
    const float4 gp = dnew_p ? *(const float4*)(dnew_p + bt * Dp + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 hv = *(const float4*)(hp + H + d);
    *(float4*)(dhp + H + d) = gate4(hv, make_float4(gp.x * cg, gp.y * cg, gp.z * cg, gp.w * cg));
    acc += gp.x * hv.x;
    acc += gp.y * hv.y;
    acc += gp.z * hv.z;
    acc += gp.w * hv.w;
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
  CSG_LAUNCH(k_embed_fwd, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, idx, rows, idx_stride, table, num_emb,
                     dim, out, out_stride, out_off);
  return check_launch("csg_embed_fwd");
}

int64_t csg_embed_bwd_workspace(int64_t rows, int64_t num_emb, int64_t dim) {
  if (rows <= 0 || num_emb <= 0 || dim <= 0) return 0;
  const int64_t chunks = cdiv(rows, emb_chunk_rows(dim));
  return chunks > 1 ? chunks * num_emb * dim * (int64_t)sizeof(float) : 0;
}

int csg_embed_bwd(const int64_t* idx, int64_t rows, int64_t idx_stride, const float* dout, int64_t out_stride,
                  int64_t out_off, int64_t num_emb, int64_t dim, float* dtable, float* workspace, int64_t workspace_bytes,
                  void* stream) {
  CSG_REQUIRE(rows >= 0 && dim > 0 && dim <= 256 && num_emb > 0 && num_emb * dim < (1ll << 30), CSG_E_BADSHAPE,
              "csg_embed_bwd: bad shape");
  if (rows == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_EMBED_BWD, (double)rows * dim * 8, s);
  const int chunk = emb_chunk_rows(dim);
  const int64_t chunks = cdiv(rows, chunk), n = num_emb * dim;
  CSG_REQUIRE(chunks < (1ll << 31) && cdiv(n, 256) < 65536, CSG_E_UNSUPPORTED, "csg_embed_bwd: too many rows / entries");
  const dim3 grid((unsigned)chunks, (unsigned)cdiv(n, 256));
  const size_t shm = ((size_t)chunk * dim + chunk) * sizeof(float);
  if (chunks == 1) {
    CSG_LAUNCH(k_embed_bwd, grid, dim3(256), shm, s, idx, rows, idx_stride, dout, out_stride, out_off, (int)num_emb, (int)dim,
               chunk, dtable, (int64_t)0, 1);
    return check_launch("csg_embed_bwd");
  }
  CSG_REQUIRE(workspace != nullptr && workspace_bytes >= chunks * n * (int64_t)sizeof(float), CSG_E_WORKSPACE,
              "csg_embed_bwd: workspace %ld < %ld bytes", (long)workspace_bytes, (long)(chunks * n * sizeof(float)));
  CSG_LAUNCH(k_embed_bwd, grid, dim3(256), shm, s, idx, rows, idx_stride, dout, out_stride, out_off, (int)num_emb, (int)dim,
             chunk, workspace, n, 0);
  int rc = check_launch("csg_embed_bwd");
  if (rc) return rc;
  CSG_LAUNCH(k_embed_bwd_sum, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, workspace, (int)chunks, (int)n, dtable);
  return check_launch("csg_embed_bwd(sum)");
}

int csg_real_object_mask(const int64_t* objs, int64_t B, int64_t O, int64_t A, int64_t image_id, uint8_t* mask,
                         void* stream) {
  CSG_REQUIRE(B >= 0 && O >= 0 && A > 0, CSG_E_BADSHAPE, "csg_real_object_mask: bad shape");
  if (B * O == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_OBJ_MASK, (double)B * O * 9, s);
  CSG_LAUNCH(k_obj_mask, dim3((unsigned)cdiv(B * O, 256)), dim3(256), 0, s, objs, B * O, A, image_id, mask);
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
  const size_t shm = csr_sorted_lds(T, O);
  if (O <= 254 && T >= 512 && T <= 65535 && shm <= 150 * 1024) {      // dense graphs: counting sort
    static size_t attr = 0;
    if (shm > attr) {
      (void)hipFuncSetAttribute((const void*)k_csr_build_sorted, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
      attr = shm;
    }
    CSG_LAUNCH(k_csr_build_sorted, dim3((unsigned)B), dim3(256), shm, s, triplets, (int)T, (int)O, row_ptr,
                       col);
  } else {
    CSG_LAUNCH(k_csr_build, dim3((unsigned)B), dim3(256), 0, s, triplets, (int)T, (int)O, row_ptr, col);
  }
  return check_launch("csg_graph_csr_build");
}

int csg_gather_concat_fwd(const float* obj, const float* pred, const int64_t* triplets, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* out, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && Din > 0 && Dp > 0, CSG_E_BADSHAPE, "csg_gather_concat_fwd: bad shape");
  CSG_REQUIRE(Din % 4 == 0 && Dp % 4 == 0 && ((uintptr_t)obj % 16) == 0 && ((uintptr_t)pred % 16) == 0 &&
                  ((uintptr_t)out % 16) == 0,
              CSG_E_BADSHAPE, "csg_gather_concat_fwd: Din=%ld and Dp=%ld must be multiples of 4, pointers 16-byte aligned",
              (long)Din, (long)Dp);
  if (T == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t n = B * T * (2 * Din + Dp);
  ProfScope p(K_GATHER_FWD, (double)n * 8, s);
  CSG_LAUNCH(k_gather_concat_fwd, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, s, obj, pred, triplets, B * T,
                     (int)O, (int)T, (int)Din, (int)Dp, out);
  return check_launch("csg_gather_concat_fwd");
}

int64_t csg_gather_concat_bwd_workspace(int64_t B, int64_t O, int64_t T, int64_t Din) {
  if (B <= 0 || O <= 0 || T < 0 || Din <= 0) return -1;
  return rowsum_ws_bytes(B, O, T, Din, false);
}

int csg_gather_concat_bwd(const float* dcat, const int32_t* row_ptr, const int32_t* col, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* dobj, float* dpred, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && Din > 0 && Dp > 0, CSG_E_BADSHAPE, "csg_gather_concat_bwd: bad shape");
  CSG_REQUIRE(Din % 4 == 0 && Dp % 4 == 0, CSG_E_BADSHAPE,
              "csg_gather_concat_bwd: Din=%ld and Dp=%ld must be multiples of 4", (long)Din, (long)Dp);
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: dcat read once, dobj (O x Din) and dpred (T x Dp) written
  ProfScope p(K_GATHER_BWD, (double)B * (T * (2.0 * Din + Dp) * 4 + O * Din * 4.0 + T * Dp * 4.0), s);
  if (dobj) {
    const int64_t ns = rowsum_nseg(O, T);
    const int64_t need = rowsum_ws_bytes(B, O, T, Din, false);
    CSG_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), CSG_E_BADSHAPE,
                "csg_gather_concat_bwd: workspace too small (%ld bytes, need %ld)", (long)workspace_bytes, (long)need);
    float* part = (float*)workspace;
    int32_t* seg_info = ns ? (int32_t*)(part + B * ns * Din) : nullptr;
    CSG_LAUNCH(k_csr_rowsum<false>, dim3((unsigned)(ns ? ns : O), (unsigned)B), dim3(256), 0, s, dcat,
                       (const float*)nullptr, (const uint8_t*)nullptr, row_ptr, col, (int)O, (int)T, (int)Din,
                       (int)(2 * Din + Dp), 0, (int)(Din + Dp), rowsum_lpe(Din), (int)ns, rowsum_seg(), dobj, (float*)nullptr, part,
                       (float*)nullptr, seg_info);
    if (ns)
      CSG_LAUNCH(k_rowsum_finish<false>, dim3((unsigned)(B * O)), dim3(1024), 0, s, (const float*)part,
                         (const float*)nullptr, (const int32_t*)seg_info, (int)O, (int)Din, rowsum_lpe(Din), (int)ns, dobj,
                         (float*)nullptr);
  }
  if (dpred && T > 0) {
    int64_t n = B * T * Dp;
    CSG_LAUNCH(k_slice_copy, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, s, dcat, B * T,
                       (int)(2 * Din + Dp), (int)Din, (int)Dp, dpred);
  }
  return check_launch("csg_gather_concat_bwd");
}

int64_t csg_segment_avg_fwd_workspace(int64_t B, int64_t O, int64_t T, int64_t H) {
  if (B <= 0 || O <= 0 || T < 0 || H <= 0) return -1;
  return rowsum_ws_bytes(B, O, T, H, true);
}

int csg_segment_avg_fwd(const float* h, const float* conf, const uint8_t* valid, const int32_t* row_ptr,
                        const int32_t* col, int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, float* pooled,
                        float* cnt, float* new_p, void* workspace, int64_t workspace_bytes, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && H > 0 && Dp >= 0, CSG_E_BADSHAPE, "csg_segment_avg_fwd: bad shape");
  CSG_REQUIRE(H % 4 == 0 && Dp % 4 == 0, CSG_E_BADSHAPE, "csg_segment_avg_fwd: H=%ld and Dp=%ld must be multiples of 4",
              (long)H, (long)Dp);
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: messages 2*T*H*4 + indices + confidence, pooled O*H*4 written
  ProfScope p(K_SEGAVG_FWD, (double)B * (T * (2.0 * H * 4 + 16 + 4) + O * H * 4.0), s);
  const int64_t ns = rowsum_nseg(O, T);
  const int64_t need = rowsum_ws_bytes(B, O, T, H, true);
  CSG_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), CSG_E_BADSHAPE,
              "csg_segment_avg_fwd: workspace too small (%ld bytes, need %ld)", (long)workspace_bytes, (long)need);
  float* part = (float*)workspace;
  float* part_cnt = ns ? part + B * ns * H : nullptr;
  int32_t* seg_info = ns ? (int32_t*)(part_cnt + B * ns) : nullptr;
  CSG_LAUNCH(k_csr_rowsum<true>, dim3((unsigned)(ns ? ns : O), (unsigned)B), dim3(256), 0, s, h, conf, valid,
                     row_ptr, col, (int)O, (int)T, (int)H, (int)(2 * H + Dp), 0, (int)(H + Dp), rowsum_lpe(H), (int)ns,
                     rowsum_seg(), pooled, cnt, part, part_cnt, seg_info);
  if (ns)
    CSG_LAUNCH(k_rowsum_finish<true>, dim3((unsigned)(B * O)), dim3(1024), 0, s, (const float*)part,
                       (const float*)part_cnt, (const int32_t*)seg_info, (int)O, (int)H, rowsum_lpe(H), (int)ns, pooled,
                       cnt);
  if (new_p && Dp > 0 && T > 0) {
    int64_t n = B * T * Dp;
    CSG_LAUNCH(k_scale_slice, dim3((unsigned)cdiv(n / 4, 256)), dim3(256), 0, s, h, conf, B * T,
                       (int)(2 * H + Dp), (int)H, (int)Dp, new_p);
  }
  return check_launch("csg_segment_avg_fwd");
}

int csg_segment_avg_bwd(const float* dpooled, const float* dnew_p, const float* h, const float* conf,
                        const uint8_t* valid, const int64_t* triplets, const float* pooled, const float* cnt,
                        int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, int32_t gate_relu, float* dh,
                        float* dconf, float* dcnt_scratch, void* stream) {
  CSG_REQUIRE(B > 0 && O > 0 && T >= 0 && H > 0 && Dp >= 0, CSG_E_BADSHAPE, "csg_segment_avg_bwd: bad shape");
  CSG_REQUIRE(H % 4 == 0 && Dp % 4 == 0, CSG_E_BADSHAPE, "csg_segment_avg_bwd: H=%ld and Dp=%ld must be multiples of 4", (long)H,
              (long)Dp);
  CSG_REQUIRE((((uintptr_t)dpooled | (uintptr_t)h | (uintptr_t)dh | (uintptr_t)dnew_p) & 15) == 0, CSG_E_UNSUPPORTED,
              "csg_segment_avg_bwd: pointers must be 16-byte aligned");
  if (T == 0) return CSG_OK;
  hipStream_t s = (hipStream_t)stream;
  ProfScope p(K_SEGAVG_BWD, (double)B * T * (2.0 * H + Dp) * 8, s);   // h read + dh written (dpooled rows are L2 hits)
  CSG_LAUNCH(k_segavg_dcnt, dim3((unsigned)(B * O)), dim3(64), 0, s, dpooled, pooled, cnt, (int)H,
                     dcnt_scratch);
  CSG_LAUNCH(k_segment_avg_bwd, dim3((unsigned)(B * T)), dim3(64), 0, s, dpooled, dnew_p, h, conf, valid,
                     triplets, cnt, dcnt_scratch, (int)O, (int)T, (int)H, (int)Dp, (int)gate_relu, dh, dconf);
  return check_launch("csg_segment_avg_bwd");
}

}  // extern "C"
