#!/usr/bin/env python3
"""Where one G+D step spends its time (config C3, B=16, 256x256): synchronised wall time per phase
and the per-kernel sums inside each phase.  Development tool."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from canonicalsg2im_amd import _lib, train as T  # noqa: E402
from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab  # noqa: E402


def main():
    use_img_disc = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dev = torch.device("cuda:0")
    vocab = make_vocab("coco")
    cfg = BASELINE_CONFIGS["C3"]["cfg"]
    opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--use_img_disc", str(use_img_disc),
                             "--batch_size", "16"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    batch = [None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(16, 256, cfg.min_objects,
                                                                                       cfg.max_objects, cfg.graph), 1)]
    for _ in range(2):
        tr.step(batch)
    torch.cuda.synchronize()
    phases = {}

    def mark(name, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        prof = _lib.prof_read()
        _lib.prof_reset()
        ksum = sum(v[0] for v in prof.values())
        top = sorted(prof.items(), key=lambda kv: -kv[1][0])[:4]
        phases.setdefault(name, []).append(((t1 - t0) * 1e3, ksum, top))
        return time.perf_counter()

    _lib.prof_reset()
    _lib.prof_enable(True)
    for _ in range(3):
        t = time.perf_counter()
        imgs, objs, boxes, triplets, _, tt, masks, _ = batch
        out = tr.model(objs, triplets, tt, boxes_gt=boxes, masks_gt=masks)
        t = mark("1 model forward (GCN + generator)", t)
        tr._d_requires_grad(False)
        G = tr.gans_model(batch, out, mode="compute_generator_loss")
        t = mark("2 G losses (D passes on fake+real)", t)
        tr.optimizer.zero_grad(set_to_none=True)
        G["total_loss"].mean().backward()
        t = mark("3 G backward", t)
        tr.optimizer.step()
        tr._d_requires_grad(True)
        t = mark("4 G Adam", t)
        D = tr.gans_model(batch, out, mode="compute_discriminator_loss")
        t = mark("5 D losses (forward passes)", t)
        tr.discriminator.optimizer_d_img.zero_grad(set_to_none=True)
        D["total_img_loss"].backward()
        if not use_img_disc:
            tr.discriminator.optimizer_d_obj.zero_grad(set_to_none=True)
            D["total_obj_loss"].backward()
        t = mark("6 D backward", t)
        tr.discriminator.optimizer_d_img.step()
        if not use_img_disc:
            tr.discriminator.optimizer_d_obj.step()
        t = mark("7 D Adam", t)
    _lib.prof_enable(False)
    tot = 0.0
    for name, rows in phases.items():
        wall = sum(r[0] for r in rows[1:]) / (len(rows) - 1)
        ks = sum(r[1] for r in rows[1:]) / (len(rows) - 1)
        tot += wall
        top = ", ".join("%s %.1f" % (k, v[0]) for k, v in rows[-1][2])
        print("%-40s wall %7.2f ms   csg kernels %7.2f ms   [%s]" % (name, wall, ks, top))
    print("sum of phases %.2f ms" % tot)


if __name__ == "__main__":
    main()
