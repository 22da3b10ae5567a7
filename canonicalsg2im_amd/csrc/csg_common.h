// Shared host-side plumbing of libcsg_hip.so: error text, launch checking, per-kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/csg_hip.h"

namespace csg {

// kernel ids for the profiling table (order = csg_prof_kernel_name)
enum KernelId {
  K_EMBED_FWD = 0,
  K_EMBED_BWD,
  K_OBJ_MASK,
  K_CSR_BUILD,
  K_GATHER_FWD,
  K_GATHER_BWD,
  K_SEGAVG_FWD,
  K_SEGAVG_BWD,
  K_LAYOUT_FWD,
  K_LAYOUT_BWD,
  K_IGEMM_FWD,    // k_igemm_fwd<128>
  K_IGEMM_FWD64,  // k_igemm_fwd<64>
  K_SPLITK_EPI,
  K_IGEMM_WGRAD,
  K_WGRAD_REDUCE,
  K_ACT_BWD,
  K_COLSUM,
  K_NORM_STATS,
  K_NORM_FINALIZE,
  K_NORM_APPLY_FWD,
  K_NORM_BWD_REDUCE,
  K_NORM_BWD_DX,
  K_UPSAMPLE_FWD,
  K_UPSAMPLE_BWD,
  K_AVGPOOL_FWD,
  K_AVGPOOL_BWD,
  K_CROP_FWD,
  K_CROP_BWD,
  K_MAXPOOL_FWD,
  K_MAXPOOL_BWD,
  K_L1_FWD,
  K_L1_BWD,
  K_CANON_BUILD,
  K_CANON_EMIT,
  K_SPECTRAL_FWD,
  K_SPECTRAL_BWD,
  K_WINO_CONV,
  K_WINO_PACK,
  K_WINO_WGRAD,
  K_FEW_FWD,
  K_FEW_BWD_DATA,
  K_FEW_BWD_WEIGHT,
  K_WINO4_CONV,
  K_WINO4_WGRAD,
  K_COUNT
};

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// profiling hooks (csg_api.hip)
bool prof_on(int kid);
void prof_begin(int kid, double work, hipStream_t s);
void prof_end(hipStream_t s);

struct ProfScope {
  hipStream_t s;
  bool on;
  ProfScope(int kid, double work, hipStream_t st) : s(st), on(prof_on(kid)) {
    if (on) prof_begin(kid, work, s);
  }
  ~ProfScope() {
    if (on) prof_end(s);
  }
};

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace csg

#define CSG_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      csg::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)
