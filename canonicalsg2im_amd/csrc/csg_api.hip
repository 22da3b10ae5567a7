// Version, error text and the per-kernel timing table of libcsg_hip.so.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "csg_common.h"

namespace csg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return CSG_E_LAUNCH;
  }
  return CSG_OK;
}

// ---- timing: start / stop events bound to the kernel's own dispatch (hipExtLaunchKernel), resolved lazily in
// csg_prof_read
struct Rec {
  int kid;
  double work;          // the scope's work rides on its first launch
  hipEvent_t e0, e1;
  bool first;           // first launch of its scope: counts as one call of the entry point
};
static int g_prof = 0;   // 0 off, 1 every kernel, 2 only the dominant convolution kernels (k_wino*_conv, k_igemm_fwd<128>), 3 only the streaming (HBM-bound) kernels
static std::mutex g_mu;
static std::vector<Rec> g_pending;
static std::vector<hipEvent_t> g_free;
static double g_ms[K_COUNT];
static double g_work[K_COUNT];
static int64_t g_n[K_COUNT];

static bool hbm_kernel(int kid) {
  switch (kid) {
    case K_GATHER_FWD: case K_GATHER_BWD: case K_SEGAVG_FWD: case K_SEGAVG_BWD: case K_LAYOUT_FWD: case K_LAYOUT_BWD:
    case K_ACT_BWD: case K_NORM_STATS: case K_NORM_APPLY_FWD: case K_NORM_BWD_REDUCE: case K_NORM_BWD_DX:
    case K_UPSAMPLE_FWD: case K_UPSAMPLE_BWD: case K_AVGPOOL_FWD: case K_AVGPOOL_BWD:
      return true;
    default:
      return false;
  }
}

bool prof_on(int kid) {
  return g_prof == 1 || (g_prof == 2 && (kid == K_IGEMM_FWD || kid == K_WINO_CONV || kid == K_WINO4_CONV)) ||
         (g_prof == 3 && hbm_kernel(kid));
}

bool prof_serialize() { return g_prof == 1 || g_prof == 3; }

ProfCur& prof_cur() {
  static thread_local ProfCur cur = {0, 0.0, false, false};
  return cur;
}

static hipEvent_t get_event() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_free.empty()) {
    hipEvent_t e = g_free.back();
    g_free.pop_back();
    return e;
  }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void prof_begin(int kid, double work) {
  ProfCur& c = prof_cur();
  c.kid = kid;
  c.work = work;
  c.armed = true;
  c.used = false;
}

void prof_next(hipEvent_t& e0, hipEvent_t& e1) {
  ProfCur& c = prof_cur();
  e0 = get_event();
  e1 = get_event();
  std::lock_guard<std::mutex> lk(g_mu);
  g_pending.push_back(Rec{c.kid, c.used ? 0.0 : c.work, e0, e1, !c.used});
  c.used = true;
}

void prof_end() {
  ProfCur& c = prof_cur();
  c.armed = c.used = false;
}

static void resolve() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_pending) {
    hipEventSynchronize(r.e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.e0, r.e1);
    g_ms[r.kid] += ms;
    g_work[r.kid] += r.work;
    g_n[r.kid] += r.first ? 1 : 0;
    g_free.push_back(r.e0);
    g_free.push_back(r.e1);
  }
  g_pending.clear();
}

static const char* kNames[K_COUNT] = {
    "embed_fwd",      "embed_bwd",      "real_object_mask", "graph_csr_build", "gather_concat_fwd", "gather_concat_bwd",
    "segment_avg_fwd", "segment_avg_bwd", "layout_fwd",       "layout_bwd",      "igemm_fwd",         "igemm_fwd64",   "splitk_epilogue", "igemm_wgrad",
    "wgrad_reduce",   "act_bwd",        "colsum",           "norm_stats",      "norm_finalize",     "norm_apply_fwd",
    "norm_bwd_reduce", "norm_bwd_dx",    "upsample2x_fwd",   "upsample2x_bwd",  "avgpool3s2_fwd",    "avgpool3s2_bwd", "crop_fwd", "crop_bwd",
    "maxpool2_fwd",   "maxpool2_bwd",   "l1_mean_fwd",      "l1_mean_bwd",
    "canon_build",    "canon_emit",     "spectral_norm_fwd", "spectral_norm_bwd",
    "wino_conv",      "wino_pack",      "wino_wgrad",
    "few_fwd",        "few_bwd_data",   "few_bwd_weight",
    "wino4_conv",     "wino4_wgrad",
    "gemm_nt",        "gemm_tn"};

}  // namespace csg

extern "C" {

int csg_version(void) { return 108; }
const char* csg_last_error(void) { return csg::g_err; }

int csg_prof_enable(int on) {
  csg::resolve();
  csg::g_prof = on < 0 ? 0 : (on > 3 ? 1 : on);
  return CSG_OK;
}

int csg_prof_reset(void) {
  csg::resolve();
  memset(csg::g_ms, 0, sizeof(csg::g_ms));
  memset(csg::g_work, 0, sizeof(csg::g_work));
  memset(csg::g_n, 0, sizeof(csg::g_n));
  return CSG_OK;
}

int csg_prof_num_kernels(void) { return csg::K_COUNT; }

const char* csg_prof_kernel_name(int kid) {
  if (kid < 0 || kid >= csg::K_COUNT) return "";
  return csg::kNames[kid];
}

int csg_prof_read(int kid, double* ms, int64_t* launches, double* work) {
  CSG_REQUIRE(kid >= 0 && kid < csg::K_COUNT, CSG_E_BADSHAPE, "csg_prof_read: bad kernel id %d", kid);
  csg::resolve();
  if (ms) *ms = csg::g_ms[kid];
  if (launches) *launches = csg::g_n[kid];
  if (work) *work = csg::g_work[kid];
  return CSG_OK;
}

}  // extern "C"
