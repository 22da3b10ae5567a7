"""Developer aid: csg_gemm_nt alone on the graph encoder's shapes (raw C ABI), TFLOP/s.  CSG_GEMM_CFG=322|162 and CSG_GEMM_TN_ROWS=32|16 select the variants."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from canonicalsg2im_amd._lib import GemmDesc, lib, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
for M, K, N in [(96000, 384, 512), (96000, 512, 1152), (96000, 1408, 512), (96000, 512, 512), (96000, 1152, 512), (1048576, 128, 64)]:
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev)
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    d = GemmDesc()
    d.M, d.N, d.K, d.lda, d.ldb, d.ldy, d.ldg, d.act, d.slope, d.gate_slope = M, N, K, K, K, N, N, 1, 0.0, 0.0
    for _ in range(3):
        lib.csg_gemm_nt(d, ptr(a), ptr(w), ptr(b), None, ptr(y), stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.csg_gemm_nt(d, ptr(a), ptr(w), ptr(b), None, ptr(y), stream())
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print("nt cfg %s  M %7d K %4d N %4d  %.3f ms  %.1f TFLOP/s" % (os.environ.get("CSG_GEMM_CFG", "default"), M, K, N, t, 2.0 * M * K * N / t / 1e9))

for M, N, K in [(96000, 512, 384), (96000, 1152, 512), (96000, 512, 1408), (96000, 512, 512), (96000, 128, 512), (1048576, 64, 128)]:
    dy = torch.randn(M, N, device=dev)
    x = torch.randn(M, K, device=dev)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    nbytes = lib.csg_gemm_tn_workspace(M, N, K)
    ws = torch.empty(max(nbytes // 4, 4), device=dev)

    def run():
        lib.csg_gemm_tn(M, N, K, ptr(dy), N, ptr(x), K, ptr(dw), ptr(db), ptr(ws), nbytes, stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print("tn rows %s  M %7d N %4d K %4d  %.3f ms  %.1f TFLOP/s  (workspace %d MB)" % (os.environ.get("CSG_GEMM_TN_ROWS", "16"), M, N, K, t,
                                                                                 2.0 * M * K * N / t / 1e9, nbytes >> 20))
