"""Developer aid: the graph encoder's linears (config C5 shapes) and the 1x1 shortcuts, per direction: the GEMM kernels of
csrc/gemm.hip, the implicit-GEMM convolution kernel (CSG_GEMM=off), and rocBLAS through torch as the yardstick.
    python tools/gemm_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from canonicalsg2im_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M0 = int(os.environ.get("M", "96000"))
shapes = [(M0, 384, 512), (M0, 512, 1152), (M0, 1408, 512), (M0, 512, 512), (M0, 512, 128),
          (16 * 256 * 256, 128, 64), (16 * 128 * 128, 256, 128), (16 * 64 * 64, 512, 256), (16 * 32 * 32, 1024, 512)]


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


def three(x, w, b, g):
    """(forward, backward-data, weight gradient) times in ms through ops.linear."""
    y = ops.linear(x, w, b, ops.ACT_LEAKY, 0.0, grad_is_pre=True)
    t_f = timeit(lambda: ops.linear(x, w, b, ops.ACT_LEAKY, 0.0, grad_is_pre=True))
    t_dx = timeit(lambda: torch.autograd.grad(y, [x], g, retain_graph=True))
    t_dw = timeit(lambda: torch.autograd.grad(y, [w, b], g, retain_graph=True))
    return t_f, t_dx, t_dw


print("%8s %5s %5s | %-32s | %-32s | %-32s" % ("M", "K", "N", "gemm.hip fwd / dx / dw  TFLOP/s", "igemm.hip", "rocBLAS"))
for M, K, N in shapes:
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = torch.randn(N, K, device=dev, requires_grad=True)
    b = torch.randn(N, device=dev, requires_grad=True)
    g = torch.randn(M, N, device=dev)
    fl = 2.0 * M * K * N / 1e9
    ops.GEMM_MODE = "all"
    a = three(x, w, b, g)
    ops.GEMM_MODE = "off"
    c = three(x, w, b, g)
    ops.GEMM_MODE = "auto"
    xd, wd = x.detach(), w.detach()
    r = (timeit(lambda: torch.addmm(b.detach(), xd, wd.t())), timeit(lambda: torch.mm(g, wd)), timeit(lambda: torch.mm(g.t(), xd)))
    print("%8d %5d %5d | %s | %s | %s" % (M, K, N, "  ".join("%6.1f" % (fl / t) for t in a), "  ".join("%6.1f" % (fl / t) for t in c),
                                       "  ".join("%6.1f" % (fl / t) for t in r)))
