"""Winograd weight packs (k_wino4_pack<3>, k_wino_pack) at the generator's weight shapes: us per launch in a tight loop and
inside a replayed HIP graph of 64 back-to-back packs (what a training step pays)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402

SHAPES = [(128, 32), (256, 128), (512, 128), (1024, 128), (2048, 128), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 1024)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("%-14s %8s %8s %8s %8s | graph of 64: us per pack" % ("Cout x Cin", "F4 fwd", "F4 bwd", "F2 fwd", "F2 bwd"))
for Cout, Cin in SHAPES:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    row = [timed(lambda: ops.wino_pack(w, bd, None, var)) for var in (4, 2) for bd in (False, True)]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.wino_pack(w, False, None, 4)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            keep = [ops.wino_pack(w, i & 1 == 1, None, 4) for i in range(64)]
    us = timed(g.replay, 20) / 64
    print("%5d x %-6d %8.1f %8.1f %8.1f %8.1f | %8.1f" % (Cout, Cin, row[0], row[1], row[2], row[3], us), flush=True)
