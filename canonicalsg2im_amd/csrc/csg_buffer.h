// Raw buffer loads through a buffer descriptor (SRD): a lane whose element does not exist simply gets an offset at
// or beyond num_records and the hardware returns zeros — loaders stay branch-free, so hipcc can interleave them
// with the MFMAs of the current tile (exec-masked loads split the loop into basic blocks and idle the matrix pipe).
// hipcc 7.2 lowers __builtin_amdgcn_raw_buffer_load_b128 to a ONE-dword load (verified on hardware, tools/probe),
// so the LLVM intrinsics are declared directly, with the descriptor as four plain dwords.
#pragma once
#include <hip/hip_runtime.h>

typedef float csg_f32x4 __attribute__((ext_vector_type(4)));
typedef float csg_f32x2 __attribute__((ext_vector_type(2)));
typedef int csg_i32x4 __attribute__((ext_vector_type(4)));

__device__ csg_f32x4 csg_buf_load_x4(csg_i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v4f32");
__device__ csg_f32x2 csg_buf_load_x2(csg_i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.v2f32");
__device__ float csg_buf_load_x1(csg_i32x4 rsrc, int voffset, int soffset, int aux) __asm(
    "llvm.amdgcn.raw.buffer.load.f32");

#define CSG_OOB_OFF 0x80000000u          // >= any num_records we ever set: the load returns 0
#define CSG_MAX_RECORDS 0x7FFFFFF0ll

__device__ __forceinline__ csg_i32x4 csg_make_srd(const void* base, long long bytes) {
  const unsigned long long a = (unsigned long long)base;
  csg_i32x4 r;
  r.x = (int)(a & 0xffffffffu);
  r.y = (int)((a >> 32) & 0xffffu);   // stride 0: raw buffer, offsets in bytes
  r.z = (int)(bytes < CSG_MAX_RECORDS ? bytes : CSG_MAX_RECORDS);
  r.w = 0x00020000;                   // DATA_FORMAT_32
  return r;
}
