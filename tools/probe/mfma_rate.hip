// Matrix-pipe rates on gfx950, registers only: v_mfma_f32_32x32x2_f32 against v_mfma_f32_32x32x16_bf16, alone and with
// packed-fp32 VALU work of the same wave between the MFMAs (does the VALU hide under the bf16 matrix pipe the way it does
// NOT hide under the fp32 one?).  Groundwork for the bf16 x 3 operand split of DESIGN.md section 7.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int VALU>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const float a = seed + threadIdx.x * 1e-6f, b = seed * 0.5f;
  bf16x8 ah, bh;
  for (int e = 0; e < 8; ++e) {
    ah[e] = (__bf16)(a + e);
    bh[e] = (__bf16)(b - e);
  }
  f32x2 v[8];
  for (int e = 0; e < 8; ++e) v[e] = f32x2{a + e, b - e};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (KIND == 0)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      else
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < VALU; ++u) v[(i * VALU + u) & 7] = __builtin_elementwise_fma(v[(i * VALU + u) & 7], f32x2{1.0001f, 0.9999f}, f32x2{1e-7f, -1e-7f});
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e][0] + v[e][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int VALU>
static void run(const char* name, float* out) {
  const int blocks = 256 * 8, iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_rate<KIND, VALU>), dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_rate<KIND, VALU>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 /*waves*/ * iters * 4;
  const double flop = mfma * 2.0 * 32 * 32 * (KIND == 0 ? 2 : 16);
  printf("%-44s %8.3f ms  %8.1f TFLOP/s   %6.1f ns per MFMA per SIMD-slot\n", name, ms, flop / ms / 1e9,
         ms * 1e6 / (mfma / (256.0 * 4)));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0, 0>("fp32 32x32x2", out);
  run<0, 2>("fp32 32x32x2 + 2 v_pk_fma per MFMA", out);
  run<0, 8>("fp32 32x32x2 + 8 v_pk_fma per MFMA", out);
  run<1, 0>("bf16 32x32x16", out);
  run<1, 2>("bf16 32x32x16 + 2 v_pk_fma per MFMA", out);
  run<1, 4>("bf16 32x32x16 + 4 v_pk_fma per MFMA", out);
  run<1, 8>("bf16 32x32x16 + 8 v_pk_fma per MFMA", out);
  return 0;
}
