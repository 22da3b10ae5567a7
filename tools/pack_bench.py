"""k_wino_pack per weight shape (channels-last parameters, forward and backward-data operands): us and GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402

SHAPES = [(32, 128), (128, 128), (128, 256), (128, 512), (128, 1024), (128, 2048), (64, 64), (128, 64), (256, 128),
          (256, 256), (512, 256), (512, 512), (1024, 512), (1024, 1024)]
for Cin, Cout in SHAPES:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    row = []
    for bwd in (False, True):
        for _ in range(3):
            ops.wino_pack(w, bwd)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.wino_pack(w, bwd)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append("%s %7.1f us %6.0f GB/s" % ("dgrad" if bwd else "fwd  ", us, Cin * Cout * 4 * 25 / us / 1e3))
    print("Cin %4d Cout %4d  %s | %s" % (Cin, Cout, row[0], row[1]), flush=True)
