#!/usr/bin/env python3
"""Host-side enqueue time of one training step vs its GPU time (is the step CPU-bound?).  On config C3 the host needs
~34 ms to enqueue a 158 ms step; later steps return in ~150 ms only because the object discriminator's index event
makes the host wait for the previous step's work.  Usage (GPU box): python tools/host_overhead.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.build()
from canonicalsg2im_amd import train as T
from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
base = BASELINE_CONFIGS["C3"]; vocab = make_vocab(base["vocab"]); cfg = base["cfg"]
dev = torch.device("cuda:0")
opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--batch_size", "16"])
torch.manual_seed(0)
tr = T.Trainer(opt, dev)
batch = [None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(16, 256, cfg.min_objects, cfg.max_objects, cfg.graph), seed=1)]
for _ in range(3): tr.step(batch)
torch.cuda.synchronize()
host = []
t_all0 = time.perf_counter()
for _ in range(6):
    t0 = time.perf_counter(); tr.step(batch); host.append(time.perf_counter() - t0)
t_enq = time.perf_counter() - t_all0
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all0
print("host enqueue per step (ms):", [round(h * 1e3, 1) for h in host], "total enqueue", round(t_enq * 1e3, 1), "total incl. drain", round(t_all * 1e3, 1))
