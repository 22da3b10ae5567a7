"""Developer aid: the graph encoder's linears (config C5 shapes) on the implicit-GEMM kernels, per direction.
    python tools/gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from canonicalsg2im_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "96000"))
shapes = [(384, 512), (512, 1152), (1408, 512), (512, 512), (512, 128)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for K, N in shapes:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = torch.randn(N, K, device=dev, requires_grad=True)
    b = torch.randn(N, device=dev, requires_grad=True)
    g = torch.randn(M, N, device=dev)
    fl = 2.0 * M * K * N
    t_f = timeit(lambda: ops.linear(x, w, b, ops.ACT_LEAKY, 0.0))
    y = ops.linear(x, w, b, ops.ACT_LEAKY, 0.0)

    def fb():
        x.grad = w.grad = b.grad = None
        yy = ops.linear(x, w, b, ops.ACT_LEAKY, 0.0)
        yy.backward(g)
    t_fb = timeit(fb)
    t_mm = timeit(lambda: torch.addmm(b, x, w.t()))
    print("M %d K %4d N %4d | fwd %.3f ms %.1f TF | fwd+bwd %.3f ms (bwd %.1f TF avg over dx+dw+act) | rocBLAS fwd %.3f ms %.1f TF"
          % (M, K, N, t_f, fl / t_f / 1e9, t_fb, 2 * fl / max(t_fb - t_f, 1e-9) / 1e9, t_mm, fl / t_mm / 1e9))
