// Shared host-side plumbing of libcsg_hip.so: error text, launch checking, per-kernel timing.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <tuple>
#include <utility>

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
  K_GEMM_NT,
  K_GEMM_TN,
  K_COUNT
};

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- per-kernel timing (csg_api.hip).  A ProfScope arms ONE launch: the next CSG_LAUNCH on this thread goes out through
// hipExtLaunchKernel with a start and a stop event bound to the kernel's own dispatch packet, so the pair reads the
// kernel's begin and end timestamps — what rocprofv3 reads.  The first implementation bracketed the launch with two
// hipEventRecord markers: those are barrier packets with a system-scope release/acquire, which write back and invalidate
// the L2 between a producer and its consumer; k_norm_apply_fwd, which normally reads the gamma/beta map its producer
// left in L2, ran 1.7x slower under them (profiles/README.md, "what an event pair times").
bool prof_on(int kid);
struct ProfCur {
  int kid;
  double work;
  bool armed, used;
};
ProfCur& prof_cur();
void prof_begin(int kid, double work);
void prof_end();
// the event pair of the next launch of the armed scope (every launch of a scope is timed: an entry point may be several
// kernels — partial sums + their ordered reduction, a split launch + its slab sum — and all of them are its time)
void prof_next(hipEvent_t& e0, hipEvent_t& e1);

bool prof_serialize();   // modes 1 and 3 (the untimed per-kernel table): drain the stream before the timed launch
struct ProfScope {
  bool on;
  ProfScope(int kid, double work, hipStream_t s) : on(prof_on(kid)) {
    if (on) {
      // A dispatch's start timestamp is taken when the command processor picks the packet up, which can be while the
      // previous kernel of the stream is still draining: a short consumer right behind a long producer (k_norm_apply_fwd
      // behind the gamma/beta convolution, k_splitk_epilogue behind its GEMM) then reads ~2x its own duration.
      // rocprofv3 serialises dispatches and does not see this; the table modes do the same here.
      if (prof_serialize()) (void)hipStreamSynchronize(s);
      prof_begin(kid, work);
    }
  }
  ~ProfScope() {
    if (on) prof_end();
  }
};

template <size_t... I, typename... KArgs>
static inline void launch_ext_impl(const void* fn, dim3 g, dim3 b, size_t shm, hipStream_t s, hipEvent_t e0, hipEvent_t e1,
                                   std::tuple<KArgs...>& stored, std::index_sequence<I...>) {
  void* ptrs[] = {(void*)&std::get<I>(stored)...};
  (void)hipExtLaunchKernel(fn, g, b, ptrs, shm, s, e0, e1, 0);
}

// kernel<<<g, b, shm, s>>>(args...) — or, when a ProfScope is armed on this thread, the same launch carrying its events
template <typename... KArgs, typename... Args>
static inline void launch(void (*kernel)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, Args&&... args) {
  ProfCur& pc = prof_cur();
  if (pc.armed) {
    if (pc.used && prof_serialize()) (void)hipStreamSynchronize(s);     // as in ProfScope: start stamps at pickup
    hipEvent_t e0, e1;
    prof_next(e0, e1);
    std::tuple<KArgs...> stored(static_cast<KArgs>(std::forward<Args>(args))...);
    launch_ext_impl((const void*)kernel, g, b, shm, s, e0, e1, stored, std::index_sequence_for<KArgs...>{});
  } else {
    hipLaunchKernelGGL(kernel, g, b, shm, s, static_cast<KArgs>(std::forward<Args>(args))...);
  }
}

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace csg

#define CSG_LAUNCH(kernel, grid, block, shm, stream, ...) csg::launch(kernel, grid, block, shm, stream, __VA_ARGS__)

#define CSG_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      csg::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

// Per-device "already done" flags for idempotent one-time host setup (hipFuncSetAttribute on a kernel): lock-free and
// re-entrant — two threads racing on a device's first launch both do the setup (harmless), a device id beyond the table does
// it on every launch (correct, a few microseconds slower).
struct DeviceOnce {
  std::atomic<bool> done[32];
  bool pending(int dev) const { return dev < 0 || dev >= 32 || !done[dev].load(std::memory_order_acquire); }
  void mark(int dev) {
    if (dev >= 0 && dev < 32) done[dev].store(true, std::memory_order_release);
  }
};
