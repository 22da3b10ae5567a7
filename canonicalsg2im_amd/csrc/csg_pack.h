// Multi-tensor Winograd weight packs (csg_wino_pack_weights_multi): up to CSG_PACK_MULTI weights per launch, the item
// table passed by value in the kernel arguments (capturable in a HIP graph, no device-side table to keep alive).
// A pack of one weight is ~7 us of fixed latency whatever its size (tools/pack_bench.py); a training step of the
// generator needs ~100 of them.
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

#define CSG_PACK_MULTI 24

struct PackMultiItem {
  const float* w;          // (Cout,Cin,3,3), element strides below, already in (n, k) roles
  float4* up;              // packed operand
  int s_n, s_k, s_h, s_w;
  int flip, N, K, NT32, Q8;
  int start;               // first block of this item in the launch
};

struct PackMulti {
  PackMultiItem it[CSG_PACK_MULTI];
  int n;
};

// F(2x2,3x3) items (wino.hip) / F(4x4,3x3) items (wino4.hip): one launch of `blocks` blocks
// (library-internal: hidden from the C ABI)
extern "C" __attribute__((visibility("hidden"))) int csg_wino2_pack_multi_launch(const PackMulti* pm, int blocks, double bytes,
                                                                                 hipStream_t s);
extern "C" __attribute__((visibility("hidden"))) int csg_wino4_pack_multi_launch(const PackMulti* pm, int blocks, double bytes,
                                                                                 hipStream_t s);
