/*
 * csg_hip.h — C ABI of libcsg_hip.so: the hand-written gfx950 (MI355X / CDNA4) kernels of the
 * CanonicalSg2Im training hot path.
 *
 * The reference (roeiherz/CanonicalSg2Im) has no FFI of its own: its hot path is PyTorch/ATen
 * calls issued from Python.  Each entry point below therefore replaces one ATen call site of the
 * reference (file:line given per function, relative to the reference root); the Python binding
 * that a maintainer adds on the reference side is the ctypes stub shown in INTEGRATION.md
 * (the build's own binding is canonicalsg2im_amd/_lib.py).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host.
 *   - the caller allocates every buffer, including workspaces; the library keeps no pointer.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no internal sync,
 *     safe to capture in a hipGraph (profiling mode excepted).
 *   - return 0 on success, a negative CSG_E_* code otherwise; csg_last_error() has the text.
 *     The Python side raises RuntimeError — errors are never swallowed (cf. the reference's
 *     blanket try/except at scripts/train.py:354,440-441, which is NOT reproduced).
 *   - activations are NHWC fp32 ("pixels x channels", channel stride 1); a tensor may be a channel
 *     slice of a wider buffer: `*_cs` is the number of floats per pixel of the underlying buffer.
 *   - indices are int64 as in the reference's collate output (sg2im/data/packed_coco.py:467-478).
 */
#ifndef CSG_HIP_H
#define CSG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSG_OK 0
#define CSG_E_BADSHAPE (-1)
#define CSG_E_UNSUPPORTED (-2)
#define CSG_E_LAUNCH (-3)
#define CSG_E_WORKSPACE (-4)

#define CSG_ACT_NONE 0
#define CSG_ACT_LEAKY 1 /* y = x>0 ? x : slope*x ; ReLU is slope 0 */
#define CSG_ACT_TANH 2

#define CSG_MAX_TAPS 16

/* 100 + the number of additive revisions of this header: entry points are only ever added, never changed or removed
 * (108: csg_wino4_conv_spade, csg_wino4_conv_spade_supported, csg_avgpool3s2_bwd_add, csg_hinge_mean_fwd / _bwd) */
int csg_version(void);
const char* csg_last_error(void);

/* ---- per-kernel timing (HIP events on the launch stream; used by bench.py's roofline) -------
 * DEVELOPER / BENCHMARK ONLY, and the one piece of MUTABLE PROCESS-GLOBAL STATE in the library (a table of pending event
 * pairs and per-kernel sums behind a mutex, csrc/csg_api.hip): every compute entry point is stateless and re-entrant with
 * the mode off (the default) — apart from immutable once-per-device attributes (raised dynamic-LDS limits) and the
 * thread-local text of csg_last_error().  With a mode on, launches are not graph-capturable and calls from several host
 * threads share one table.
 * mode 0 = off, 1 = every launch, 2 = only the dominant convolution kernels (k_wino4_conv_v, k_wino_conv2<*>, k_igemm_fwd<128>):
 * 3 = only the streaming (HBM-bound) kernels: normalisation, activation, layout, graph gathers, resampling;
 * an event pair costs ~9 us of queue time, 10 ms per step when all ~1100 launches carry one, 2 ms in mode 2. */
int csg_prof_enable(int mode);
int csg_prof_reset(void);
int csg_prof_num_kernels(void);
const char* csg_prof_kernel_name(int kernel_id);
/* synchronises the recorded events; ms = summed launch durations, work = summed algorithmic
 * FLOPs (MFMA-bound kernels) or bytes (HBM-bound kernels) as stated in DESIGN.md */
int csg_prof_read(int kernel_id, double* ms, int64_t* launches, double* work);

/* ---- K1: attribute / predicate embedding lookup ---------------------------------------------
 * replaces nn.Embedding x A + torch.cat (sg2im/attribute_embed.py:40-45) and
 * pred_embeddings (sg2im/model.py:109).
 * out[r, out_off + 0..dim) = table[idx[r*idx_stride], :]                                       */
int csg_embed_fwd(const int64_t* idx, int64_t rows, int64_t idx_stride, const float* table, int64_t num_emb,
                  int64_t dim, float* out, int64_t out_stride, int64_t out_off, void* stream);
/* dtable[idx[r], :] += dout[r, out_off..] — replaces the backward of nn.Embedding (attribute_embed.py:40-45): rows added in
 * row order (no atomics, bit-reproducible).  More than one chunk of 8192 / dim rows needs a workspace of csg_embed_bwd_workspace(...) bytes for
 * the per-chunk partial tables (0 = none needed).
 * LIMITS: dim <= 256 and num_emb * dim < 2^30 (CSG_E_BADSHAPE otherwise; the reference's --embedding_dim is 128).  The
 * workspace is chunks * num_emb * dim floats and the grid re-stages a chunk once per 256 table entries: sized for vocabularies of
 * 10^2..10^3 rows (COCO 184, VG 179 + 46 predicates, CLEVR 4 x <= 9), not for 10^5-row tables.            */
int64_t csg_embed_bwd_workspace(int64_t rows, int64_t num_emb, int64_t dim);
int csg_embed_bwd(const int64_t* idx, int64_t rows, int64_t idx_stride, const float* dout, int64_t out_stride,
                  int64_t out_off, int64_t num_emb, int64_t dim, float* dtable, float* workspace, int64_t workspace_bytes,
                  void* stream);

/* ---- real-object mask: sg2im/utils.py:56-63 (remove_dummy_objects), batched, bit-exact ------ */
int csg_real_object_mask(const int64_t* objs, int64_t B, int64_t O, int64_t A, int64_t image_id, uint8_t* mask,
                         void* stream);

/* ---- K2/K5 support: per-image CSR of triplets by incident object ------------------------------
 * row_ptr (B,O+1) int32, col (B,2T) int32 with col = 2*t + role (role 0 subject, 1 object).
 * Inside a row: all subject entries in t order, then all object entries in t order — the order in
 * which the reference's two scatter_add calls accumulate (sg2im/graph.py:98-99).               */
int csg_graph_csr_build(const int64_t* triplets, int64_t B, int64_t T, int64_t O, int32_t* row_ptr, int32_t* col,
                        void* stream);

/* K2: cur_t = cat(obj[s], pred, obj[o])   (sg2im/graph.py:63-66) */
int csg_gather_concat_fwd(const float* obj, const float* pred, const int64_t* triplets, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* out, void* stream);
/* dobj[b,i] = sum over object i's CSR row of the matching slice of dcat; dpred = the middle slice.
 * Dense graphs split each row over several workgroups: the partial sums live in `workspace`
 * (csg_gather_concat_bwd_workspace bytes, 0 for sparse graphs) and are combined in a fixed order.
 * Din and Dp must be multiples of 4 (16-byte rows). */
int64_t csg_gather_concat_bwd_workspace(int64_t B, int64_t O, int64_t T, int64_t Din);
int csg_gather_concat_bwd(const float* dcat, const int32_t* row_ptr, const int32_t* col, int64_t B, int64_t O,
                          int64_t T, int64_t Din, int64_t Dp, float* dobj, float* dpred, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* K4+K5: confidence gate + masked segment average (sg2im/graph.py:69-109).
 * h = net1 output (B,T,2H+Dp) = [s | p | o]; conf (B,T); valid (B,T) = pred_indicators.
 * pooled (B,O,H), cnt (B,O), new_p (B,T,Dp) = conf * h[:, H:H+Dp]
 * H and Dp multiples of 4.  `workspace`: csg_segment_avg_fwd_workspace bytes (row-split partials of
 * dense graphs; 0 for sparse ones).                                                             */
int64_t csg_segment_avg_fwd_workspace(int64_t B, int64_t O, int64_t T, int64_t H);
int csg_segment_avg_fwd(const float* h, const float* conf, const uint8_t* valid, const int32_t* row_ptr,
                        const int32_t* col, int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, float* pooled,
                        float* cnt, float* new_p, void* workspace, int64_t workspace_bytes, void* stream);
/* dcnt_scratch (B,O) float workspace.  gate_relu = 1: h is the output of a ReLU (the final non-linearity of net1,
 * sg2im/graph.py:67) and dh is written as the gradient of its PRE-activation (zero where h <= 0): the producing Linear
 * needs no activation-derivative pass. */
int csg_segment_avg_bwd(const float* dpooled, const float* dnew_p, const float* h, const float* conf,
                        const uint8_t* valid, const int64_t* triplets, const float* pooled, const float* cnt,
                        int64_t B, int64_t O, int64_t T, int64_t H, int64_t Dp, int32_t gate_relu, float* dh,
                        float* dconf, float* dcnt_scratch, void* stream);

/* ---- K6: boxes_to_layout (sg2im/layout.py:12-45) and masks_to_layout (:48-77, train mode), batched
 * out[b, y, x, out_off + d] = sum_o valid[b,o] * vecs[b,o,d] * cov(y_src) * cov(x_src)
 * With `masks` (B,O,M,M) fp32 non-NULL the weight of object o at a pixel is the bilinear sample of
 * its mask over the box (grid_sample, align_corners=False, zeros padding) instead of cov*cov.
 * (OH,OW) may be smaller than (H,W): output pixel y samples full-resolution row
 * floor(y*H/OH) — the nearest resize of generator.py:99 / normalization.py:102 folded in.       */
int csg_layout_fwd(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                   int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                   int64_t out_cs, int64_t out_off, void* stream);
/* The image discriminator's input `torch.cat([img, layout], dim=1)` (spade/models/networks/discriminator.py:120) built
 * in one pass at full resolution: out (B,H,W,out_cs) NHWC with channels [layout(S) | img(3) | zeros], out_cs >= S + 4 a
 * multiple of 4 (the first convolution's weight is permuted to that order by the caller).  `img` (B,3,H,W) fp32 with
 * element strides (img_sb, img_sc, img_sh, img_sw) — contiguous or channels-last alike.  The layout channels are exactly
 * csg_layout_fwd's.                                                                                                    */
int csg_disc_input_fwd(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M, int64_t B,
                       int64_t O, int64_t S, int64_t H, int64_t W, const float* img, int64_t img_sb, int64_t img_sc,
                       int64_t img_sh, int64_t img_sw, float* out, int64_t out_cs, void* stream);
/* dvecs (B,O,S) = (accumulate ? dvecs : 0) + sum_{y,x} dout * cov * cov.  With `dboxes` (B,O,4) non-NULL (needs
 * `vecs`) the gradient w.r.t. [x0,y0,w,h] is produced too: the grid of layout.py:98-110 is differentiable in the
 * box, and grid_sample's backward w.r.t. its grid is the bilinear weights' derivative.
 * `workspace` (csg_layout_bwd_workspace bytes; 0 when the shape is not served): boxes_to_layout without box gradients
 * on maps from 32 rows up runs as two passes that read dout ONCE — per-tile partial sums for the objects active in the
 * tile, then an ordered sum per object (bit-reproducible).  Without it, one block per (object, image) walks the
 * object's own box support (dout is read once per covering object).                                          */
int64_t csg_layout_bwd_workspace(int64_t B, int64_t O, int64_t S, int64_t OH, int64_t OW, int32_t has_masks,
                                 int32_t box_gradients);
int csg_layout_bwd(const float* dout, int64_t out_cs, int64_t out_off, const float* boxes, const uint8_t* valid,
                   const float* masks, int64_t M, int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH,
                   int64_t OW, float* dvecs, int accumulate, const float* vecs, float* dboxes, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* dmasks (B,O,M,M) = (accumulate ? dmasks : 0) + the gradient of masks_to_layout (layout.py:48-77) w.r.t. the masks:
 * grid_sample's backward w.r.t. its input, summed over the embedding channels (dout . vecs per pixel).  One block per
 * (object, image), every mask cell an ordered sum.  Rows of invalid objects are zero.                         */
int csg_layout_bwd_masks(const float* dout, int64_t out_cs, int64_t out_off, const float* boxes, const uint8_t* valid,
                         int64_t M, int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW,
                         const float* vecs, float* dmasks, int accumulate, void* stream);
/* masks_to_layout(test_mode=True) (layout.py:71-74,135-151): painter's compositing, one object per pixel.
 * csg_layout_mass: mass[b,o] = sum(samples[o]) at full resolution (+inf for invalid objects) — the caller sorts it
 * (ascending, stable) into `order` (B,O) int32, -1 after the last valid object.
 * csg_layout_paint: pixel -> first object of `order` whose sampled mask is > 0.5, value vec * mask sample.  */
int csg_layout_mass(const float* vecs, const float* boxes, const uint8_t* valid, const float* masks, int64_t M,
                    int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, float* mass, void* stream);
int csg_layout_paint(const float* vecs, const float* boxes, const float* masks, int64_t M, const int32_t* order,
                     int64_t B, int64_t O, int64_t S, int64_t H, int64_t W, int64_t OH, int64_t OW, float* out,
                     int64_t out_cs, int64_t out_off, void* stream);

/* ---- K3/K8/K11: implicit-GEMM convolution on fp32 MFMA ---------------------------------------
 * replaces nn.Conv2d (generator.py:28,46; architecture.py:29-32; normalization.py:89-94;
 * discriminator.py:175-187) and nn.Linear (sg2im/layers.py:10; a Linear is a 1x1 conv on an
 * (M,1,1,K) image).  One descriptor covers forward, backward-data (transposed taps, weights packed
 * [Cin][tap][Cout]) and the parity classes of a stride-2 transposed convolution.
 *
 * GEMM view: rows m = (b, gy, gx) over the output grid, cols n = output channel,
 * k = (tap slot, input channel).  Virtual input coordinate of tap t for grid point (gy,gx):
 *   iy = gy*istride + tap_dy[t], ix = gx*istride + tap_dx[t]; outside [0,IHv)x[0,IWv) reads 0;
 *   the physical source pixel is (iy >> in_up, ix >> in_up) (nearest 2x upsample folded in).
 * Packed weights: w[n][tap_w[t]][c], c fastest, row length wtaps*Cin.
 * Output pixel of grid point: (gy*os + ooy, gx*os + oox) in an (OHf,OWf) image.
 * res_gate = 1: `residual` is not added but GATES the result, y = result * (residual > 0 ? 1 : slope) — the derivative
 * of the (Leaky)ReLU whose output `residual` is: a backward-data pass hands the producer of its input the gradient of the
 * PRE-activation and no separate activation-derivative pass runs.  Requires act == CSG_ACT_NONE (slope is then the
 * gate's negative slope).                                                                                            */
typedef struct csg_conv_desc {
  int32_t B, IHp, IWp, Cin, x_cs;
  int32_t IHv, IWv, in_up;
  int32_t OHg, OWg, OHf, OWf, os, ooy, oox;
  int32_t Cout, y_cs;
  int32_t istride, ntaps, wtaps;
  int32_t tap_dy[CSG_MAX_TAPS], tap_dx[CSG_MAX_TAPS], tap_w[CSG_MAX_TAPS];
  int32_t act;
  float slope;
  int32_t accumulate; /* y += result (after bias/act) instead of y = */
  int32_t res_gate; /* 1: the residual gates the result instead of being added (see above) */
} csg_conv_desc;

/* y = act(conv(x, w) + bias) [+ residual];  bias and residual may be NULL; residual has y's layout.
 * Layers whose output grid is too small to fill 256 CUs are split along K into `workspace` slabs
 * that an ordered second pass sums (csg_conv_fwd_workspace bytes; a NULL/short workspace when that is non-zero
 * is CSG_E_WORKSPACE, as for the multi and weight-gradient entry points). */
int64_t csg_conv_fwd_workspace(const csg_conv_desc* d);
int csg_conv_fwd(const csg_conv_desc* d, const float* x, const float* w, const float* bias, const float* residual,
                 float* y, float* workspace, int64_t workspace_bytes, void* stream);
/* Up to four descriptors that share weights, input, output tensor, channel counts and batch, served by ONE launch: the
 * parity classes of a stride-2 transposed convolution (backward-data of the PatchGAN's 4x4/2 layers,
 * discriminator.py:175-183).  Each class alone fills a fraction of the chip.  Workspace as csg_conv_fwd's.        */
int64_t csg_conv_fwd_multi_workspace(const csg_conv_desc* descs, int32_t n);
int csg_conv_fwd_multi(const csg_conv_desc* descs, int32_t n, const float* x, const float* w, const float* bias,
                       const float* residual, float* y, float* workspace, int64_t workspace_bytes, void* stream);
/* dw[n][tap][c] = sum_m dy[m][n] * x[src(m,tap)][c]; `d` is the FORWARD descriptor (y_cs = floats per
 * pixel of dy).  Deterministic split-K: partial slabs in `workspace`, then an ordered reduction.
 * db (Cout floats, may be NULL) receives the bias gradient sum_m dy[m][n], accumulated from the dY
 * tiles the kernel stages anyway (no separate pass over dy). */
int64_t csg_conv_bwd_weight_workspace(const csg_conv_desc* d);
int csg_conv_bwd_weight(const csg_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                        float* workspace, int64_t workspace_bytes, void* stream);

/* ---- K8w: 3x3 / stride 1 / pad 1 convolution by Winograd F(2x2,3x3) on fp32 MFMA (csrc/wino.hip) ----
 * The same nn.Conv2d call sites as above restricted to 3x3 kernels (generator.py:28; architecture.py:29-31;
 * normalization.py:89-94) and their backward-data passes: 16 instead of 36 multiplications per 2x2 output tile
 * and (cin,cout) pair, fp32 throughout.  x (B,H,W,Cin) NHWC with channel stride x_cs, y likewise with y_cs;
 * H, W even; channel counts multiples of 4.  y = act(conv + bias) [+ residual] exactly as csg_conv_fwd.
 * `packed`: the transformed weights U = G g G^T in MFMA operand order, csg_wino_pack_bytes(N, K) bytes, produced by
 * csg_wino_pack_weights from the (Cout,Cin,3,3) weight with element strides (s_o,s_i,s_h,s_w) — contiguous or
 * channels-last parameters alike: backward_data = 0 -> operand of the forward
 * (N = Cout, K = Cin); 1 -> operand of dX = conv(dY, flipped W^T) (N = Cin, K = Cout: call csg_wino_conv with
 * Cin := Cout, Cout := Cin).  `sigma` (device scalar or NULL) divides every weight first — W / sigma of spectral
 * normalisation — so a spectrally normalised layer needs no materialised W_eff.                            */
typedef struct csg_wino_desc {
  int32_t B, H, W;
  int32_t Cin, x_cs;
  int32_t Cout, y_cs;
  int32_t act;
  float slope;
} csg_wino_desc;
int64_t csg_wino_pack_bytes(int64_t N, int64_t K);
int csg_wino_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                          int32_t backward_data, const float* sigma, float* packed, void* stream);
/* `workspace` (csg_wino_conv_workspace(d) bytes, may be 0 / NULL): slabs of a split over the input channels, used
 * when the tile grid alone cannot fill the chip and the call has no epilogue (bias, residual, activation) — the
 * backward-data pass of the 128 -> 2048 gamma/beta convolutions; summed in a fixed order (bit-reproducible).
 * Without (enough) workspace the launch runs unsplit.
 * `gate` (nullable, the layout of y): y *= (gate > 0 ? 1 : gate_slope) as the last step — the derivative of the
 * (Leaky)ReLU that PRODUCED this convolution's input, folded into the backward-data pass of the SPADE gamma/beta
 * convolutions (their input `actv` = ReLU(mlp_shared(seg)), normalization.py:96-100, has no other consumer), which
 * replaces a separate pass over the 128-channel maps.  A gated call is never split.                              */
int64_t csg_wino_conv_workspace(const csg_wino_desc* d);
int csg_wino_conv(const csg_wino_desc* d, const float* x, const float* packed, const float* bias,
                  const float* residual, const float* gate, float gate_slope, float* y, float* workspace,
                  int64_t workspace_bytes, void* stream);
/* Weight gradient of the same layers by Winograd F(3x3,2x2): dw [Cout][3][3][Cin] (the layout of
 * csg_conv_bwd_weight), db (Cout) or NULL; x (B,H,W,Cin) the layer's input, dy (B,H,W,Cout) the gradient of its
 * pre-activation output.  Deterministic: per-slice slabs in `workspace` + an ordered reduction.              */
int64_t csg_wino_bwd_weight_workspace(const csg_wino_desc* d);
int csg_wino_bwd_weight(const csg_wino_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                        int64_t workspace_bytes, void* stream);

/* Weight gradient of the same layers by Winograd F(3x3,4x4) (csrc/wino4w.hip): 36 multiplications per 4x4 tile of dY where
 * F(3x3,2x2) needs 64 — the layers of architecture.py:29-31 and normalization.py:89-94 on maps whose H and W are multiples
 * of 8 (efficient from 64 input and 64 output channels up: a block owns 64 x 64 channels).  Same arguments, layouts and
 * determinism as csg_wino_bwd_weight; csg_wino4_bwd_weight_workspace < 0: the shape is not served.                   */
int64_t csg_wino4_bwd_weight_workspace(const csg_wino_desc* d);
int csg_wino4_bwd_weight(const csg_wino_desc* d, const float* x, const float* dy, float* dw, float* db, float* workspace,
                         int64_t workspace_bytes, void* stream);

/* ---- K8w4: the same 3x3 / stride 1 / pad 1 layers by Winograd F(4x4,3x3) (csrc/wino4.hip) ------------------------
 * 36 multiplications per 4x4 output tile and (cin,cout) pair — 2.25 per output where F(2x2,3x3) needs 4 — on maps at
 * least 32 pixels wide (H, W multiples of 4, H >= 16, Cin a multiple of 8); interpolation points {0, 1, -1, 1/2, -2,
 * inf}, error against fp64 < 1e-5 of the output scale (tests/test_gpu_wino4.py).  Same descriptor, arguments and
 * epilogue as csg_wino_conv; the packed operand has 36 positions (csg_wino4_pack_bytes / csg_wino4_pack_weights, same
 * arguments as the F(2x2,3x3) pack).  csg_wino4_supported(d) = 1 when a layer is served (0 also when CSG_WINO4=0).   */
int32_t csg_wino4_supported(const csg_wino_desc* d);
/* Several (Cout,Cin,3,3) weights packed by ONE launch per kernel family and 24 items: a pack costs ~7 us of fixed latency
 * whatever its size (tools/pack_bench.py) and a generator step needs about a hundred.  Each item is what
 * csg_wino_pack_weights (variant 2, F(2x2,3x3)) / csg_wino4_pack_weights (variant 4, F(4x4,3x3)) take, without the
 * spectral-norm divisor; outputs are bit-identical to the one-weight calls.  `items` is host memory, read during the
 * call (the table travels in the kernel arguments: the launch can be captured in a HIP graph).                        */
typedef struct csg_wino_pack_item {
  const float* w;            /* (Cout,Cin,3,3) with element strides s_o, s_i, s_h, s_w */
  int64_t s_o, s_i, s_h, s_w;
  int64_t Cout, Cin;
  int32_t backward_data;     /* 1: the operand of the backward-data pass (channel roles swapped, taps flipped) */
  int32_t variant;           /* 2 | 4 */
  float* packed;             /* csg_wino_pack_bytes / csg_wino4_pack_bytes (N, K) bytes, 16-byte aligned */
} csg_wino_pack_item;
int csg_wino_pack_weights_multi(const csg_wino_pack_item* items, int32_t n, void* stream);
/* Launches with at least two (region, 64-channel block) items per CU, an even number of 8-channel stages (>= 4) and
 * Cout a multiple of 64 run as ONE block per CU that walks its items with the stage pipeline carried across them
 * (bit-identical outputs).  csg_wino4_persistent(0 | 1) switches that form off / on for the process
 * and returns the previous setting (-1 = not yet decided: CSG_WINO4_PERSIST, default 1); a negative argument only
 * queries.  For A/B measurements and the parity tests — not needed in normal use.                                    */
int32_t csg_wino4_persistent(int32_t on);
int64_t csg_wino4_pack_bytes(int64_t N, int64_t K);
int csg_wino4_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                           int32_t backward_data, const float* sigma, float* packed, void* stream);
int64_t csg_wino4_conv_workspace(const csg_wino_desc* d);
int csg_wino4_conv(const csg_wino_desc* d, const float* x, const float* packed, const float* bias,
                   const float* residual, const float* gate, float gate_slope, float* y, float* workspace,
                   int64_t workspace_bytes, void* stream);

/* A slice of the outputs of an F(4x4,3x3) convolution — 32-channel tiles [tile_off, tile_off + Cout/32) of a packed
 * operand of `tiles_total` tiles — optionally with the SPADE modulation (normalization.py:96-110) as its epilogue:
 * with mod_x != NULL the launch is the BETA half of the gamma || beta convolution and writes
 *     y = leaky(((mod_x - mean) * invstd) * (1 + gamma) + (conv + bias), mod_slope)
 * where `mod_gamma` (pixel stride gamma_cs) is the gamma map a plain launch of this entry point (tile_off 0, y_cs =
 * gamma_cs) wrote before, mod_x has y's layout and mean / invstd are the (C,) batch statistics.  beta never reaches
 * memory and the apply pass of csg_norm_apply_fwd is not needed.  `bias` points at the slice's first channel.      */
int csg_wino4_conv_part(const csg_wino_desc* d, const float* x, const float* packed, int64_t tile_off, int64_t tiles_total,
                        const float* bias, const float* mod_x, const float* mod_gamma, int64_t gamma_cs,
                        const float* mod_mean, const float* mod_invstd, float mod_slope, float* y, void* stream);

/* ONE launch for a SPADE modulation (normalization.py:89-110; round 6): the gamma || beta convolution (`packed`: the ordinary
 * F(4x4,3x3) forward operand of the (2C, Cin, 3, 3) weight, gamma tiles then beta tiles; `bias`: its 2C biases) with
 *     y = leaky(((mod_x - mean) * invstd) * (1 + gamma) + beta, mod_slope)
 * as the epilogue of blocks that own a gamma tile and the beta tile of the same 32 channels.  d->Cout = C (the modulated map's
 * channels, a multiple of 32), y and mod_x are (B,H,W,y_cs).  `gamma_out` (nullable; pixel stride gamma_cs): gamma = conv +
 * bias is also written there — the backward reads it (csg_norm_apply_bwd_reduce / _dx); inference passes NULL.  beta never
 * reaches memory.  Outputs are bit-identical to csg_wino4_conv_part(gamma) followed by csg_wino4_conv_part(beta).
 * Served by the persistent form of the kernel only: csg_wino4_conv_spade_supported(d) = 1 when the launch has at least two
 * (region, 32-channel gamma + beta block) items per CU and an even number (>= 4) of 8-channel stages; else the launch pair. */
int32_t csg_wino4_conv_spade_supported(const csg_wino_desc* d);
int csg_wino4_conv_spade(const csg_wino_desc* d, const float* x, const float* packed, const float* bias, const float* mod_x,
                         float* gamma_out, int64_t gamma_cs, const float* mod_mean, const float* mod_invstd, float mod_slope,
                         float* y, void* stream);

/* ---- K8w34: 4x4 / stride 1 convolutions by Winograd F(3x3,4x4) (csrc/wino4.hip, the same kernel on 3x3 output tiles)
 * The PatchGAN's fourth layer (discriminator.py:184-189: 4x4, stride 1, padding 2, 256 -> 512 channels at 1/8
 * resolution) and its backward-data pass (the 4x4 correlation with the flipped, transposed weight and padding 1):
 * 36 multiplications per 3x3 output tile instead of 144.  d->H, d->W are the INPUT size; `pad` is 2 (output (H+1) x
 * (W+1)) or 1 (output (H-1) x (W-1)); x (B,H,W,x_cs), y (B,Ho,Wo,y_cs), residual / gate in y's layout.  Same points,
 * input transform and positions as F(4x4,3x3): the packed operand has csg_wino4_pack_bytes bytes.                    */
int32_t csg_wino34_supported(const csg_wino_desc* d, int32_t pad);
int csg_wino34_pack_weights(const float* w, int64_t s_o, int64_t s_i, int64_t s_h, int64_t s_w, int64_t Cout, int64_t Cin,
                            int32_t backward_data, const float* sigma, float* packed, void* stream);
/* `workspace` (csg_wino34_conv_workspace bytes, may be 0 / NULL) as csg_wino_conv's: slabs of a split over the input
 * channels when the tile grid alone cannot fill the chip (the half-resolution scale's backward-data pass: 128 blocks)
 * and the call has no epilogue; summed in a fixed order.                                                        */
int64_t csg_wino34_conv_workspace(const csg_wino_desc* d, int32_t pad);
int csg_wino34_conv(const csg_wino_desc* d, int32_t pad, const float* x, const float* packed, const float* bias,
                    const float* residual, const float* gate, float gate_slope, float* y, float* workspace,
                    int64_t workspace_bytes, void* stream);

/* ---- K8n: stride-1 convolutions with at most four output channels (csrc/fewn.hip) --------------------------------
 * `conv_img` (generator.py:46,120-121: 64 -> 3, 3x3, pad 1, tanh behind it) and the PatchGAN prediction heads
 * (discriminator.py:185-187: 512 -> 1, 4x4, pad 2): a 32-wide MFMA tile wastes 29 (31) of its columns on them; these
 * are VALU kernels that move the many-channel side once.  Output channels are padded to 4 (y, dy: (B,OH,OW,4); w, dw:
 * [4][KH][KW][Cin]; the forward and backward-data passes read only the first cout_real rows of w and entries of bias,
 * the weight gradient writes all four rows of dw / entries of db, zeros beyond cout_real); KH = KW in {3, 4}; Cin in {32, 64, ..., 1024};
 * KH*KW*cout_real <= 36.  The weight gradient is bit-reproducible (per-block slabs + ordered sum).               */
typedef struct csg_few_desc {
  int32_t B, IH, IW;
  int32_t Cin, x_cs;
  int32_t KH, KW, pad;
  int32_t cout_real;
  int32_t act;
  float slope;
  int32_t in_act; /* 1: the convolution sees leaky(x, in_slope): the LeakyReLU in front of conv_img (generator.py:123-124) is
                   * applied in the forward and weight-gradient loaders instead of a pass of its own; backward-data callers
                   * multiply by its derivative themselves (csg_conv_desc.res_gate) */
  float in_slope;
} csg_few_desc;
int csg_conv_few_supported(const csg_few_desc* d);      /* 1 / 0 */
/* forward `workspace` (csg_conv_few_fwd_workspace bytes, may be 0): slabs of a split over the input channels for maps
 * with few pixels and many channels (the heads); summed in a fixed order.  Without it the launch runs unsplit.    */
int64_t csg_conv_few_fwd_workspace(const csg_few_desc* d);
int csg_conv_few_fwd(const csg_few_desc* d, const float* x, const float* w, const float* bias, float* y, float* workspace,
                     int64_t workspace_bytes, void* stream);
int csg_conv_few_bwd_data(const csg_few_desc* d, const float* dy, const float* w, float* dx, void* stream);
int64_t csg_conv_few_bwd_weight_workspace(const csg_few_desc* d);
int csg_conv_few_bwd_weight(const csg_few_desc* d, const float* x, const float* dy, float* dw, float* db,
                            float* workspace, int64_t workspace_bytes, void* stream);

/* ---- plain fp32 GEMMs (csrc/gemm.hip) ------------------------------------------------------------
 * The layers of the path that ARE matrix products: nn.Linear of the graph encoder (sg2im/graph.py:63-77 through
 * sg2im/layers.py:114-140 build_mlp; rows = triplets or objects) and 1x1 convolutions on NHWC maps (SPADEResnetBlock.conv_s,
 * spade/models/networks/architecture.py:37-39; rows = pixels).  csg_conv_fwd / csg_conv_bwd_weight serve them as well (the
 * small ones stay there); these kernels drop the convolution's tap table and stage both operands by LDS-DMA.
 *   csg_gemm_nt:  y[m][n] = epi(sum_k a[m][k] * bw[n][k] + bias[n]);  epi = act, then `gate` (nullable, rows of ldg
 *                 floats): y *= (gate[m][n] > 0 ? 1 : gate_slope) — the derivative of the (Leaky)ReLU that produced this
 *                 layer's input, for the backward-data pass (a = dy, bw = W^T stored [K_layer][N_layer]).
 *   csg_gemm_tn:  dw[n][k] = sum_m dy[m][n] * x[m][k]  ([Cout][Cin] = the layout of csg_conv_bwd_weight for a 1x1
 *                 filter), db[n] = sum_m dy[m][n] (nullable).  Rows are cut into slices with one slab each in `workspace`
 *                 (csg_gemm_tn_workspace bytes, 0 = none needed), summed in a fixed order: bit-reproducible.
 * K, N (tn), every row stride: multiples of 4 floats; pointers 16-byte aligned.                                       */
typedef struct csg_gemm_desc {
  int64_t M, N, K;
  int64_t lda, ldb, ldy, ldg; /* floats per row of a, bw, y, gate */
  int32_t act;                /* CSG_ACT_NONE / CSG_ACT_LEAKY (ReLU = slope 0) */
  float slope;
  float gate_slope;
} csg_gemm_desc;
int csg_gemm_supported(const csg_gemm_desc* d);
int csg_gemm_nt(const csg_gemm_desc* d, const float* a, const float* bw, const float* bias, const float* gate, float* y,
                void* stream);
int64_t csg_gemm_tn_workspace(int64_t M, int64_t N, int64_t K);
int csg_gemm_tn(int64_t M, int64_t N, int64_t K, const float* dy, int64_t ldy, const float* x, int64_t ldx, float* dw,
                float* db, float* workspace, int64_t workspace_bytes, void* stream);

/* dpre = dy * act'(.) evaluated from the OUTPUT y (leaky: y>0 ? 1 : slope; tanh: 1-y^2)          */
int csg_act_bwd(const float* dy, const float* y, int64_t n, int32_t act, float slope, float* dpre, void* stream);
/* out[c] = sum_rows x[r, c] over (rows, C) with row stride x_cs — bias gradients; partial (nchunk,2C) fp64 */
int csg_colsum(const float* x, int64_t rows, int64_t C, int64_t x_cs, float* out, double* partial, int64_t nchunk,
               void* stream);

/* ---- K9/K11: BatchNorm / InstanceNorm statistics + SPADE modulation + LeakyReLU ---------------
 * replaces F.batch_norm (sync_batchnorm/batchnorm.py:65-68), the modulation of
 * normalization.py:108, F.leaky_relu (architecture.py:53-54,67-68) and nn.InstanceNorm2d +
 * LeakyReLU (normalization.py:44, discriminator.py:181-185).
 * x is (G groups, P pixels, C channels): BatchNorm G=1, P=B*h*w; InstanceNorm G=B, P=h*w.
 * sums (G,2C) double = [sum x | sum x^2]; it is the message SyncBN all-reduces
 * (batchnorm.py:74-83).  Accumulated in fp64 throughout (partial: (G,nchunk,2C) doubles): x and x*x are
 * exact in fp64, so E[x^2]-E[x]^2 loses nothing to the chunking.                                 */
int csg_norm_stats(const float* x, int64_t G, int64_t P, int64_t C, double* sums, double* partial, int64_t nchunk,
                   void* stream);
/* mode 0: invstd = 1/sqrt(var+eps) (F.batch_norm); mode 1: invstd = clamp(var,eps)^-1/2
 * (batchnorm.py:145, N-replica path).  running_* may be NULL; running_var gets the unbiased var. */
int csg_norm_finalize(const double* sums, int64_t G, int64_t C, double count, float eps, int32_t mode, float* mean,
                      float* invstd, float* running_mean, float* running_var, float momentum, void* stream);
/* One-rank statistics in two launches instead of three: csg_norm_stats's first stage, then ONE kernel that reduces the
 * partial table (the same fixed order: bit-identical sums) and finalises as csg_norm_finalize does with mode 0 — mean,
 * invstd = (var + eps)^-1/2 and the running statistics of up to two modules that see the same batch (norm_0 and norm_s of
 * a SPADE residual block, architecture.py:37-47).  Not for N > 1 ranks: there the sums travel between the two steps. */
int csg_norm_stats_finalize(const float* x, int64_t G, int64_t P, int64_t C, double* partial, int64_t nchunk, double count,
                            float eps, float* mean, float* invstd, float* running_mean, float* running_var,
                            float* running_mean2, float* running_var2, float momentum, void* stream);
/* y = leaky((x-mean)*invstd*(1+gamma)+beta, slope); gb (G*P, 2C) = [gamma | beta] or NULL; slope 1 = no act.
 * (gb2, slope2, y2), nullable: a second modulation of the same normalised x written in the same pass.           */
int csg_norm_apply_fwd(const float* x, const float* mean, const float* invstd, const float* gb, float slope,
                       int64_t G, int64_t P, int64_t C, float* y, const float* gb2, float slope2, float* y2,
                       void* stream);
/* pass 1: dgb (if gb; always (G*P, 2C) = [d gamma | d beta]) and dsums (G,2C) double = [sum dn | sum dn*xhat].  `yact`
 * (nullable, with gb): the activated output y of the forward — the LeakyReLU gate is then read off y's sign and beta is
 * never read; gb may then be a gamma-only map.  `gb_cs`: floats per pixel of gb — 2C ([gamma | beta]) or, with yact, C
 * (what the fused forward csg_wino4_conv_part keeps: it never materialises beta).                                  */
int csg_norm_apply_bwd_reduce(const float* dy, const float* x, const float* mean, const float* invstd,
                              const float* gb, const float* yact, float slope, int64_t G, int64_t P, int64_t C, float* dgb,
                              double* dsums, double* partial, int64_t nchunk, int64_t gb_cs, void* stream);
/* pass 2: dx = invstd*(dn - dsum0/count - xhat*dsum1/count).  (dy2, gb2, slope2), nullable: a second SPADE modulation
 * of the SAME normalised x (norm_0 and norm_s of a residual block with a learned shortcut, architecture.py:37-47, see
 * the same batch statistics): dn = dn_1 + dn_2 and `dsums` holds the sum of both pass-1 reductions — one pass and one
 * dx instead of two passes and an addition.  `dgb` / `dgb2` (nullable): pass 1's output for the same modulation — its
 * d(beta) half IS dy times the activation gate, so pass 2 reads it instead of dy and beta (one map less, same bits).
 * `gb_cs`: floats per pixel of gb and gb2 — 2C, or C for gamma-only maps (then dgb / dgb2 are required).             */
int csg_norm_apply_bwd_dx(const float* dy, const float* x, const float* mean, const float* invstd, const float* gb,
                          float slope, const double* dsums, double count, int64_t G, int64_t P, int64_t C, float* dx,
                          const float* dy2, const float* gb2, float slope2, const float* dgb, const float* dgb2,
                          int64_t gb_cs, void* stream);

/* ---- K7 / pooling ------------------------------------------------------------------------------
 * nearest 2x upsample (generator.py:48,102-121) and its adjoint */
/* F.interpolate(x, size=(OH, OW), mode='nearest') of an NHWC map (B,IH,IW,C), C a multiple of 4 — the resize of the
 * segmentation map in SPADE.forward (spade/models/networks/normalization.py:98) when the caller passes a plain tensor
 * instead of the layout pyramid; the backward sums dy over the output pixels that read each input pixel, in order. */
int csg_nearest_resize_fwd(const float* x, int64_t B, int64_t IH, int64_t IW, int64_t C, int64_t OH, int64_t OW, float* y,
                           void* stream);
int csg_nearest_resize_bwd(const float* dy, int64_t B, int64_t IH, int64_t IW, int64_t C, int64_t OH, int64_t OW, float* dx,
                           void* stream);
int csg_upsample2x_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream);
int csg_upsample2x_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream);
/* F.avg_pool2d(3, stride 2, pad 1, count_include_pad=False) (discriminator.py:92-93) */
int csg_avgpool3s2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream);
int csg_avgpool3s2_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream);
/* dx = add + avgpool_bwd(dy): the gradient of a map that feeds a consumer of its own (gradient `add`, laid out like dx; may BE
 * dx) and its pooled copy — MultiscaleDiscriminator's input (discriminator.py:120-131).  Replaces autograd's separate addition. */
int csg_avgpool3s2_bwd_add(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, const float* add, float* dx,
                           void* stream);

/* ---- object crops for the object discriminator (sg2im/bilinear.py:44-94) ---------------------------
 * out[n, y, x, c] = bilinear sample (grid_sample, align_corners=False, zeros padding) of image
 * img_idx[n] over box n ([x0,y0,w,h] in [0,1]); img is NHWC with img_cs floats per pixel, C used;
 * out is (N,HH,WW,out_cs) with channels >= C written as 0.  Backward: dimg is OVERWRITTEN — every image pixel gathers the
 * contributions of the crops that sampled it, in crop order (no atomics, bit-reproducible).
 * LIMITS of csg_crop_bwd: C <= 4 and HH, WW <= 64 (CSG_E_UNSUPPORTED otherwise; the reference crops 3-channel images at
 * --crop_size 32).  The forward and csg_crop_bwd_boxes have no such limit; the Python wrapper refuses a differentiable
 * crop outside them at FORWARD time. */
int csg_crop_fwd(const float* img, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C, const float* boxes,
                 const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, float* out, int64_t out_cs, void* stream);
int csg_crop_bwd(const float* dout, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C, const float* boxes,
                 const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, int64_t out_cs, float* dimg, void* stream);
/* Gradient w.r.t. the crop boxes (N,4) [x0,y0,w,h] — the sampling grid of crop_bbox is differentiable in them
 * (sg2im/bilinear.py:83-94); img is the forward's image, dboxes (N,4) is overwritten.  One block per crop, ordered sums. */
int csg_crop_bwd_boxes(const float* dout, const float* img, int64_t B, int64_t H, int64_t W, int64_t img_cs, int64_t C,
                       const float* boxes, const int64_t* img_idx, int64_t N, int64_t HH, int64_t WW, int64_t out_cs,
                       float* dboxes, void* stream);

/* ---- perceptual loss helpers (spade/models/networks/architecture.py:93-123, loss.py:102-117) --------
 * nn.MaxPool2d(kernel 2, stride 2) of torchvision's vgg19().features (floor mode; the backward
 * routes each gradient to the first maximum of its window, as ATen does, and needs the pool's
 * input x); nn.L1Loss() between two feature maps of n floats: out[0] = mean |a - b|, and
 * da = sign(a - b) * gout[0] / n.  The workspace holds per-block fp64 partial sums. */
int csg_maxpool2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream);
int csg_maxpool2_bwd(const float* dy, const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* dx,
                     void* stream);
/* nn.AvgPool2d(2, 2) of `build_cnn(pooling='avg')` (sg2im/layers.py:88-90): mean of each 2 x 2 window, floor mode */
int csg_avgpool2_fwd(const float* x, int64_t B, int64_t H, int64_t W, int64_t C, float* y, void* stream);
int csg_avgpool2_bwd(const float* dy, int64_t B, int64_t H, int64_t W, int64_t C, float* dx, void* stream);
int64_t csg_l1_mean_workspace(int64_t n);
int csg_l1_mean_fwd(const float* a, const float* b, int64_t n, float* out, void* workspace, int64_t workspace_bytes,
                    void* stream);
int csg_l1_mean_bwd(const float* a, const float* b, const float* gout, int64_t n, float* da, void* stream);

/* ---- GAN terms on the PatchGAN's prediction maps (spade/models/networks/loss.py:60-93, gan_mode hinge / w) in one launch:
 * out[0] = (1/n) sum_i -mean(term(x_i)) over up to four scales, term = x (kind 0: the generator's term), min(x - 1, 0)
 * (kind 1: discriminator on real), min(-x - 1, 0) (kind 2: discriminator on fake).  A map is (B, 1, H, W) with element strides
 * (sb, sh, sw); the backward writes d x_i into the contiguous (B, H, W) buffer dx.  fp64 sums in a fixed order. */
typedef struct csg_hinge_item {
  const float* x;
  float* dx;                    /* backward only */
  int64_t sb, sh, sw;
  int64_t B, H, W;
} csg_hinge_item;
int csg_hinge_mean_fwd(const csg_hinge_item* items, int32_t n, int32_t kind, float* out, void* stream);
int csg_hinge_mean_bwd(const csg_hinge_item* items, int32_t n, int32_t kind, const float* gout, void* stream);

/* ---- canonical scene-graph construction of the packed datasets -----------------------------------
 * Replaces, per batch, the per-sample numpy/python pipeline of the data loader:
 * BaseDataset.add_location_triplets -> add_dummy_triplets -> add_learnt_triplets (learned_converse=0)
 * (sg2im/data/base_dataset.py:35-151, scripts/graphs_utils.py:15-100; call order
 * sg2im/data/packed_clevr_dialog.py:205-209) and the triplet padding of the collate function
 * (packed_clevr_dialog.py:309-315).  Inputs are the padded batch tensors: objs0 (B,O) int64 ids of the
 * first attribute, boxes (B,O,4) xywh fp32, centers (B,O,2) fp32 (the dataset's obj_centers), n_objs (B,)
 * int64 number of objects of each sample INCLUDING its __image__ object (rows >= n are padding).
 * pred_ids[8] = ids of __padding__, __in_image__, __below__, __above__, __left of__, __right of__,
 * __inside__, __surrounding__.  At most 256 objects per sample.
 *   csg_canon_build: relations, transitive closure and reduction as bit matrices (kept in `workspace`,
 *     csg_canon_workspace(B) bytes) and counts[b] = {original triplets, transitive triplets}.
 *   csg_canon_emit: triplets (B,T,3) int64 sorted by (s,p,o) as np.unique(axis=0) does, then the
 *     transitive extras by (p,s,o); rows past a sample's count are [0, __padding__, 0];
 *     triplet_type (B,T) int64 is 0 / 1 (ORIGINAL_EDGE / TRANSITIVE_EDGE) and 0 in the padding.
 *     T is chosen by the caller: max_b(counts[b][0] + counts[b][1]) reproduces the collate. */
int64_t csg_canon_workspace(int64_t B);
int csg_canon_build(const int64_t* objs0, const float* boxes, const float* centers, const int64_t* n_objs,
                    int64_t B, int64_t O, const int32_t* pred_ids, int64_t image_id, int include_dummies,
                    int learned_transitivity, void* workspace, int64_t workspace_bytes, int64_t* counts,
                    void* stream);
int csg_canon_emit(const int64_t* objs0, const int64_t* n_objs, int64_t B, int64_t O, const int32_t* pred_ids,
                   int64_t image_id, int include_dummies, int learned_transitivity, const void* workspace,
                   const int64_t* counts, int64_t T, int64_t* triplets, int64_t* triplet_type, void* stream);
/* learned_converse = 1 (base_dataset.py:104-107; get_edge_converse_triplets, scripts/graphs_utils.py:126-152), between
 * csg_canon_build and csg_canon_emit: one uniform number per original triplet of the six location relations (the order of
 * the reference's loops: relations by ascending predicate id, triplets by (s, o); counts[b][0] minus the __in_image__
 * dummies of sample b) picks, through `cdf` (6 x 6 float64: per relation — in the order __below__ __above__ __left of__
 * __right of__ __inside__ __surrounding__ — the cumulative distribution numpy.random.choice forms from the softmax of the
 * five candidate weights, candidates by ascending predicate id, and a zero for "none"), whether a converse edge (o, r, s)
 * joins relation r.  `uniforms` is the host's random stream, `u_off[b]` the first number of sample b.  The workspace is
 * updated in place (current graph, transitive extras, offsets), `counts` rewritten, conv_counts (B, P, P + 1) float32 —
 * zero-initialised by the caller — receives the draw counts with column P = "none".                                    */
int csg_canon_converse(const int64_t* objs0, const int64_t* n_objs, int64_t B, int64_t O, const int32_t* pred_ids,
                       int64_t image_id, int include_dummies, int learned_transitivity, void* workspace,
                       const double* cdf, const double* uniforms, const int64_t* u_off, int64_t num_preds,
                       float* conv_counts, int64_t* counts, void* stream);

/* ---- spectral normalisation of a conv weight (a13) ----------------------------------------------
 * torch.nn.utils.spectral_norm's forward pre-hook (reference call sites architecture.py:35-39,
 * normalization.py:27; SpectralNorm.compute_weight, n_power_iterations = 1, dim = 0):
 *   iterate != 0 (training): v <- normalize(W^T u); u <- normalize(W v), both buffers updated in place;
 *   sigma = u.(W v); w_eff = w / sigma.   normalize(x) = x / max(||x||, eps).
 * w is weight_orig viewed (Cout, K = Cin*KH*KW), K % 4 == 0.  u_used / v_used receive the vectors the
 * graph keeps for backward (PyTorch clones them for the same reason).
 * Backward: dw = (dweff - (sum dweff.w_eff) u v^T) / sigma; dweff is given with its element strides
 * along (Cout, Cin, KH, KW) — rows must be dense, e.g. contiguous or the [Cout][KH][KW][Cin] layout of
 * csg_conv_bwd_weight.  One workspace size serves both calls. */
int64_t csg_spectral_norm_workspace(int64_t Cout, int64_t K);
/* cl_Cin = 0: W_eff in W's memory order; cl_Cin = Cin (a multiple of 4): W is (Cout,Cin,KH,KW) row-major and W_eff is
 * written in channels-last memory [Cout][KH][KW][Cin], the convolution kernels' forward operand (no repack).      */
int csg_spectral_norm_fwd(const float* w, float* u, float* v, int64_t Cout, int64_t K, int iterate, float eps,
                          float* w_eff, int64_t cl_Cin, float* sigma, float* u_used, float* v_used, void* workspace,
                          int64_t workspace_bytes, void* stream);
int csg_spectral_norm_bwd(const float* dweff, int64_t Cout, int64_t Cin, int64_t KH, int64_t KW, int64_t s0,
                          int64_t s1, int64_t s2, int64_t s3, const float* w, const float* u_used,
                          const float* v_used, const float* sigma, float* dw, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* Multi-tensor forms: every spectrally normalised weight of a network pass in ONE launch per stage (a generator forward
 * calls the hook of architecture.py:35-39 on 18 convolutions, a PatchGAN pass on 3 per scale).  Same arithmetic and
 * bit-identical results per weight as the single-tensor calls; `workspace` per item as csg_spectral_norm_workspace. */
typedef struct csg_sn_fwd_item {
  const float* w;
  float* u;
  float* v;
  int64_t Cout, K;
  float* w_eff;
  int64_t cl_Cin;
  float* sigma;
  float* u_used;
  float* v_used;
  void* workspace;
  int64_t workspace_bytes;
} csg_sn_fwd_item;
int csg_spectral_norm_fwd_multi(const csg_sn_fwd_item* items, int32_t n, int32_t iterate, float eps, void* stream);
typedef struct csg_sn_bwd_item {
  const float* dweff;
  int64_t Cout, Cin, KH, KW, s0, s1, s2, s3;
  const float* w;
  const float* u_used;
  const float* v_used;
  const float* sigma;
  float* dw;
  void* workspace;
  int64_t workspace_bytes;
} csg_sn_bwd_item;
int csg_spectral_norm_bwd_multi(const csg_sn_bwd_item* items, int32_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CSG_HIP_H */
