"""Which captured RCCL collectives upset ProcessGroupNCCL's watchdog on this stack?  One rank, nccl backend.
usage: python tools/probe/nccl_capture_probe.py <case>     case: fwd | bwd | bwd_async | eager_after"""
import os
import sys
import time

import torch
import torch.distributed as dist

case = sys.argv[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)


class AR(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x * 2

    @staticmethod
    def backward(ctx, g):
        g = g.clone()
        if case == "bwd_async":
            w = dist.all_reduce(g, async_op=True)
            w.wait()
        else:
            dist.all_reduce(g)
        return g * 2


x = torch.ones(1024, device="cuda", requires_grad=True)
y = torch.zeros(1024, device="cuda")
dist.all_reduce(y)           # eager warm-up (creates the communicator)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if case == "fwd":
        t = x.detach() * 3
        w = dist.all_reduce(t, async_op=True)
        w.wait()
        out = t + 1
    else:
        x.grad = None
        AR.apply(x).sum().backward()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
if case == "eager_after":
    dist.all_reduce(y)
    torch.cuda.synchronize()
time.sleep(3.0)              # give the watchdog time to trip
print("case", case, "OK", float(x.grad.sum()) if x.grad is not None else float(out.sum()))
dist.destroy_process_group()
