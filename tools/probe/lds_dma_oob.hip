// Does buffer_load_dwordx4 ... lds write ZEROS for lanes whose offset is out of range, or leave LDS untouched?
// Does the scalar offset take part in the range check?   hipcc --offload-arch=gfx950 lds_dma_oob.hip -o probe && ./probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, float* y, int nbytes, int soff) {
  __shared__ float sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nbytes, 0x00020000);
  const int voff = (threadIdx.x & 1) ? 0x80000000 : threadIdx.x * 16;       // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, voff, soff, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) y[i] = sm[i];
}
int main() {
  float *x, *y, hx[1024], hy[256];
  for (int i = 0; i < 1024; ++i) hx[i] = (float)i;
  hipMalloc(&x, 4096); hipMalloc(&y, 1024);
  hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(x, y, 1024, 0);
  hipMemcpy(hy, y, 1024, hipMemcpyDeviceToHost);
  printf("lane0 %g %g | lane1 (OOB) %g %g | lane2 %g\n", hy[0], hy[1], hy[4], hy[5], hy[8]);
  k<<<1, 64>>>(x, y, 1024, 2048);     // voffset in range, voffset + soffset beyond num_records = 1024 bytes
  hipMemcpy(hy, y, 1024, hipMemcpyDeviceToHost);
  printf("soffset 2048, num_records 1024: lane0 %g (512 = soffset is NOT range-checked, 0 = it is) lane2 %g\n", hy[0], hy[8]);
  return 0;
}
