"""Developer aid: per-segment host / device time of the replayed training step (CSG_GRAPH_TIMING=1).
    python tools/graph_timing.py [C2|C3|C4]"""
import json
import os
import sys
import time

os.environ["CSG_GRAPH_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from canonicalsg2im_amd import train as T  # noqa: E402
from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "C4"
base = BASELINE_CONFIGS[cfg_name]
H = {"C2": 128}.get(cfg_name, 256)
B = {"C2": 16, "C3": 16, "C4": 4, "C5": 6}[cfg_name]
dev = torch.device("cuda:0")
vocab = make_vocab(base["vocab"])
opt = T.make_opt(vocab, ["--image_size", "%d,%d" % (H, H), "--no_vgg_loss", "--batch_size", str(B)])
torch.manual_seed(0)
tr = T.Trainer(opt, dev)
cfg = base["cfg"]
bs = [[None if t is None else t.to(dev) for t in make_batch(vocab, BatchConfig(B, H, cfg.min_objects, cfg.max_objects, cfg.graph), seed=i)]
      for i in range(4)]
for i in range(8):                       # eager, capture, the encoder's bucket seen once, its capture, replays
    tr.step(bs[i % 4])
torch.cuda.synchronize()
print(cfg_name, "graphs:", tr.graphs.captures, "sets,", tr.graphs.sg_captures, "encoder graphs,", tr.graphs.sg_replays, "encoder replays so far")
# untimed-by-marks wall clock first (marks add a device sync per step)
marks, tr.graphs.marks = tr.graphs.marks, None
t0 = time.perf_counter()
for i in range(10):
    tr.step(bs[i % 4])
torch.cuda.synchronize()
wall = 100.0 * (time.perf_counter() - t0)
tr.graphs.marks = marks
for i in range(8):
    tr.step(bs[i % 4])
rep = marks.report()
print(cfg_name, "wall %.2f ms/step |" % wall, " | ".join("%s h%.2f d%.2f" % (k, v["host_ms"], v["device_ms"]) for k, v in rep.items()))
print(json.dumps({"config": cfg_name, "wall_ms_per_step": round(wall, 2), "segments": rep,
                  "host_sum": round(sum(v["host_ms"] for v in rep.values()), 2),
                  "device_sum": round(sum(v["device_ms"] for v in rep.values()), 2)}))
