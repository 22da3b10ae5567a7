#include <hip/hip_runtime.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_probe(const float* p, float* o, int nbytes, int scale) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
  unsigned off = threadIdx.x * scale;
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
  o[threadIdx.x * 4 + 0] = __builtin_bit_cast(float, v.x);
  o[threadIdx.x * 4 + 1] = __builtin_bit_cast(float, v.y);
  o[threadIdx.x * 4 + 2] = __builtin_bit_cast(float, v.z);
  o[threadIdx.x * 4 + 3] = __builtin_bit_cast(float, v.w);
}
extern "C" int probe(const float* p, float* o, int nbytes, int scale, void* stream) {
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, p, o, nbytes, scale);
  return (int)hipGetLastError();
}
