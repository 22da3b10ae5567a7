#!/usr/bin/env python3
"""Timing of the device-side canonical graph construction (csrc/canon.hip) at the C5 sizes, with the
numpy oracle (oracle/canon.py — the checker, timed here only as the CPU baseline) beside it.
Usage (GPU box): python tools/canon_bench.py [--batch 48] [--objects 128] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=48)
    ap.add_argument("--objects", type=int, default=128)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cpu_graphs", type=int, default=4)
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import _lib
    from canonicalsg2im_amd.sg2im.data import canonical_triplets
    from canonicalsg2im_amd.synth import make_vocab
    from oracle import canon
    vocab = make_vocab("clevr")
    rng = np.random.default_rng(0)
    B, n = args.batch, args.objects
    O = n + 1
    objs = np.zeros((B, O, 4), np.int64)
    objs[:, :n] = rng.integers(1, 3, size=(B, n, 4))
    wh = rng.uniform(0.05, 0.6, size=(B, n, 2))
    xy = rng.uniform(0, 1, size=(B, n, 2)) * (1 - wh)
    boxes = -np.ones((B, O, 4), np.float32)
    boxes[:, :n] = np.concatenate([xy, wh], axis=2)
    cen = np.zeros((B, O, 2), np.float32)
    cen[:, :n] = xy + 0.5 * wh
    nn = np.full(B, O, np.int64)
    d = [torch.from_numpy(x).cuda() for x in (objs, boxes, cen, nn)]
    out = {}
    for trans in (0, 1):
        for _ in range(3):
            t, _, tt = canonical_triplets(*d, vocab, learned_transitivity=bool(trans))
        torch.cuda.synchronize()
        _lib.prof_reset(); _lib.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            t, _, tt = canonical_triplets(*d, vocab, learned_transitivity=bool(trans))
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.reps
        prof = _lib.prof_read(); _lib.prof_enable(False)
        k = {name: round(1000.0 * ms / cnt, 1) for name, (ms, cnt, _) in prof.items()}
        t0 = time.perf_counter()
        g = min(args.cpu_graphs, B)
        canon.canonical_batch(objs[:g, :, 0], boxes[:g], cen[:g], nn[:g], vocab, bool(trans), True)
        cpu = (time.perf_counter() - t0) / g
        out["learned_transitivity=%d" % trans] = {
            "triplets_per_sample": int(t.shape[1]), "wall_ms_per_batch": round(wall * 1e3, 3),
            "graphs_per_s": round(B / wall, 1), "kernel_us": k,
            "numpy_oracle_ms_per_graph": round(cpu * 1e3, 2), "numpy_oracle_graphs_per_s": round(1.0 / cpu, 2)}
    print(json.dumps({"workload": "canonical graph construction, %d samples x %d objects (+__image__)" % (B, n),
                      "reference": "python/numpy loops: ~2.5 s per graph at O=128 (SURVEY.md 8f rank 3, measured in the "
                                   "build container)", **out}))


if __name__ == "__main__":
    main()
