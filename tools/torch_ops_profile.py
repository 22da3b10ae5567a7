#!/usr/bin/env python3
"""Which ATen ops (not libcsg kernels) cost GPU time in one training step, with shapes and python call sites.
Usage (GPU box): python tools/torch_ops_profile.py > gpurun_out/torch_ops.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
    base = BASELINE_CONFIGS["C3"]
    vocab = make_vocab(base["vocab"])
    cfg = base["cfg"]
    dev = torch.device("cuda:0")
    opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", "16"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    batch = [None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(16, 256, cfg.min_objects, cfg.max_objects, cfg.graph), seed=1)]
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        tr.step(batch)
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=12)
    rows = []
    for e in ka:
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0)
        if t > 0 and e.key.startswith("aten::"):
            st = [s for s in e.stack if "canonicalsg2im_amd" in s or "torch/nn/utils" in s or "optim" in s or "autograd" in s]
            rows.append((t, e.count, e.key, str(e.input_shapes)[:110], st[:4]))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print("total self device time of ATen ops: %.2f ms" % (tot / 1e3))
    print("launch-like ATen ops: %d" % sum(r[1] for r in rows))
    for t, c, k, sh, st in rows[:110]:
        print("%8.1f us x%-4d %-28s %s\n           %s" % (t, c, k[:28], sh, " <- ".join(s.split("/")[-1][:70] for s in st)))


if __name__ == "__main__":
    main()
