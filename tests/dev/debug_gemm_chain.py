"""Developer aid: the Linear -> ReLU -> Linear chain of tests/test_gpu_gemm.py on both paths, error per gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from canonicalsg2im_amd import ops

M, K, N = 70000, 128, 128
g = torch.Generator().manual_seed(5)
x = torch.randn(M, K, generator=g)
w1 = torch.randn(N, K, generator=g) / K ** 0.5
b1 = torch.randn(N, generator=g) * 0.1
w2 = torch.randn(K, N, generator=g) / N ** 0.5
b2 = torch.randn(K, generator=g) * 0.1
gy = torch.randn(M, K, generator=g)
ref_in = [t.clone().double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
h = F.relu(F.linear(ref_in[0], ref_in[1], ref_in[2]))
ref = F.relu(F.linear(h, ref_in[3], ref_in[4]))
ref_g = torch.autograd.grad(ref, ref_in + [h], gy.double())
for enabled in (False, True):
    ops.GEMM_MODE = "all" if enabled else "off"
    dev = [t.clone().cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    hh = ops.linear(dev[0], dev[1], dev[2], ops.ACT_LEAKY, 0.0, grad_is_pre=True)
    y = ops.linear(hh, dev[3], dev[4], ops.ACT_LEAKY, 0.0, in_act=(ops.ACT_LEAKY, 0.0))
    got = torch.autograd.grad(y, dev + [hh], gy.cuda())
    print("gemm" if enabled else "igemm", "calls", ops.gemm_calls(), "y", float((y.double().cpu() - ref).abs().max()))
    for name, a, b in zip(("dx", "dw1", "db1", "dw2", "db2", "dh(pre)"), got, ref_g):
        if name == "dh(pre)":
            b = b * (h.detach() > 0)
        print("   %-8s err %.3g scale %.3g" % (name, float((a.double().cpu() - b).abs().max()), float(b.abs().max())))
