"""Bisect the watchdog failure of the replayed step under a one-rank nccl group.  PROBE_EAGER=1 interleaves an eager trainer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", CSG_DIST_FORCE="1", CSG_GRAPHS_DIST="1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_graphs as TG

cuda = torch.device("cuda:0")
argv = ["--use_img_disc", os.environ.get("PROBE_IMG_DISC", "1")]
vocab, graphed = TG._make(cuda, argv, graphs=True)
eager = TG._make(cuda, argv, graphs=False)[1] if os.environ.get("PROBE_EAGER") == "1" else None
bs = TG._batches(vocab, cuda, 2)
if os.environ.get('PROBE_AUDIT') == '1':
    from canonicalsg2im_amd import dist as D
    D.comm_reset()
for it in range(5):
    if eager is not None:
        eager.step(bs[it % 2])
    G, D = graphed.step(bs[it % 2])
    if os.environ.get("PROBE_SLEEP", "1") == "1":
        torch.cuda.synchronize()
        time.sleep(1.0)
    print("iteration", it, "ok", float(G["total_loss"]), graphed.graphs.captures, graphed.graphs.replays, flush=True)
time.sleep(2.0)
print("DONE", flush=True)
dist.destroy_process_group()
