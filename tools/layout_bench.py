"""boxes_to_layout forward / backward on dense scenes (config C5: 6 images x 64-128 objects, S = 128) and on COCO-sized
ones (C3: 16 images x <= 30 objects, S = 32): us per launch and GB/s of the algorithmic traffic (the output, once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from canonicalsg2im_amd import ops  # noqa: E402
from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab  # noqa: E402


def timeit(fn, n=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, B, S in (("C5", 6, 128), ("C3", 16, 32)):
    base = BASELINE_CONFIGS[name]
    vocab = make_vocab(base["vocab"])
    cfg = base["cfg"]
    batch = make_batch(vocab, BatchConfig(B, 256, cfg.min_objects, cfg.max_objects, cfg.graph), seed=1)
    objs, boxes = batch[1].cuda(), batch[2].cuda()
    valid = (boxes[..., 2] > 0).to(torch.uint8).contiguous()
    O = boxes.shape[1]
    for H in (256, 64):
        vecs = torch.randn(B, O, S, device="cuda", requires_grad=True)
        out = ops.layout_pyramid(vecs, boxes, valid, H, [H])[0]
        g = torch.randn_like(out)
        t_f = timeit(lambda: ops.layout_pyramid(vecs.detach(), boxes, valid, H, [H]))
        t_b = timeit(lambda: torch.autograd.grad(out, vecs, g, retain_graph=True))
        nbytes = B * H * H * S * 4
        print("%s B=%d O=%d S=%d %dx%d  fwd %7.1f us %6.0f GB/s | bwd %7.1f us %6.0f GB/s" %
              (name, B, O, S, H, H, t_f, nbytes / t_f / 1e3, t_b, nbytes / t_b / 1e3), flush=True)
