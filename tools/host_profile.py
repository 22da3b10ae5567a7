"""cProfile of the host side of one training step at config C2 (round 3: ~29 ms of enqueue per step; round 4, with the
HIP-graph replay of canonicalsg2im_amd/graphs.py on: ~10 ms)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import __graft_entry__ as ge
    ge.build()
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
    base = BASELINE_CONFIGS["C2"]
    vocab = make_vocab(base["vocab"])
    cfg = base["cfg"]
    dev = torch.device("cuda:0")
    opt = T.make_opt(vocab, ["--image_size", "128,128", "--no_vgg_loss", "--batch_size", "16"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    batch = [None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(16, 128, cfg.min_objects, cfg.max_objects, cfg.graph), seed=1)]
    for _ in range(6):                  # eager, capture of the set, the encoder's bucket seen once, its capture, replays
        tr.step(batch)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
