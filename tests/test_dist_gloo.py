"""The N>1 path on CPU: two processes over gloo (the GPU box uses the same code over RCCL).
Covers batch sharding, bucketed gradient averaging (== the reference's DataParallel gradient
reduction because per-replica losses are means over equal shards, scripts/train.py:363,391),
parameter broadcast, and the SyncBN (sum, sum^2) exchange with the N-replica formula."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from canonicalsg2im_amd import dist as D
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab, shard_batch
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and D.world_size() == world and D.rank() == rank
    torch.manual_seed(100 + rank)                       # replicas start DIFFERENT, broadcast fixes it
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    unused = torch.nn.Linear(4, 4)                       # like G's never-used repr_net: no grad, must be skipped
    D.broadcast_module(net)
    vocab = make_vocab("tiny")
    full = make_batch(vocab, BatchConfig(4, 8, 2, 4, "random"), seed=3)
    mine = shard_batch(full, rank, world)
    x = mine[2].reshape(mine[2].shape[0], -1)[:, :6]    # boxes as features
    buckets = D.GradBuckets(list(net.parameters()) + list(unused.parameters()), bucket_bytes=256)
    # first backward: buckets do not exist yet — they are built at the first synchronisation from the UNION over ranks
    # of the parameters that received a gradient (the unused module stays out)
    buckets.begin()
    loss = net(x).pow(2).mean()
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    nbytes = buckets.all_reduce_mean()
    raw = sum(p.numel() for p in net.parameters()) * 4
    # (every slot starts on a 16-byte boundary — the weight-gradient kernels write into the slots — so a bucket may carry
    # up to 12 bytes of zero padding per parameter)
    assert buckets.built and len(buckets.flats) >= 2 and nbytes == sum((p.numel() + 3) // 4 * 4 for p in net.parameters()) * 4 + 16      # (+ the late-gradient flag's 16 bytes in the last bucket)
    assert all(id(p) not in buckets.slot for p in unused.parameters())
    averaged = [p.grad.clone() for p in net.parameters()]
    allocs = buckets.allocations
    # steady state: hooks copy each gradient into its slot during the backward and launch the bucket's all-reduce when
    # its last gradient lands; `.grad` IS the slot afterwards; nothing new is allocated; same averages
    for p in net.parameters():
        p.grad = None
    buckets.begin()
    net(x).pow(2).mean().backward()
    launched_in_backward = len(buckets._works)
    buckets.flush()
    assert launched_in_backward == len(buckets.flats) - 1     # every bucket was complete before flush(); the last one — it carries
    #                                                           the "late gradient seen" flag — is launched by flush()
    assert buckets.finish() == nbytes and buckets.allocations == allocs
    for p, g in zip(net.parameters(), averaged):
        assert torch.equal(p.grad, g)
        assert p.grad.data_ptr() == buckets.slot[id(p)][1].data_ptr()
    # a backward in which a member gets no LOCAL gradient on one rank (a shard that left a branch empty): that rank
    # contributes zeros and receives the other rank's average — same collectives on both ranks, same `.grad` afterwards
    for p in net.parameters():
        p.grad = None
    buckets.begin()
    if rank == 0:
        (net[0](x).pow(2).mean()).backward()                  # rank 0: only the first Linear
    else:
        net(x).pow(2).mean().backward()
    second_local = net[2].weight.grad.clone() if rank == 1 else torch.zeros_like(net[2].weight)
    buckets.finish()
    both2 = [torch.empty_like(second_local) for _ in range(world)]
    dist.all_gather(both2, second_local)
    assert net[2].weight.grad is not None and torch.allclose(net[2].weight.grad, sum(both2) / world, rtol=1e-6, atol=1e-9)
    assert net[2].weight.grad.data_ptr() == buckets.slot[id(net[2].weight)][1].data_ptr()
    for p, g in zip(net.parameters(), local):                 # restore the first result for the checks below
        p.grad = None
    buckets.begin()
    net(x).pow(2).mean().backward()
    buckets.finish()
    # the one-shot form on a later step: all_reduce_mean() WITHOUT begin() — the hooks were not armed, so the fresh
    # gradients are moved into their slots and exchanged then (they used to be left unreduced, silently)
    for p in net.parameters():
        p.grad = None
    net(x).pow(2).mean().backward()
    assert all(p.grad.data_ptr() != buckets.slot[id(p)][1].data_ptr() for p in net.parameters())
    D.comm_reset()
    assert buckets.all_reduce_mean() == nbytes
    for p, g in zip(net.parameters(), averaged):
        assert torch.equal(p.grad, g), "one-shot all_reduce_mean() without begin()"
        assert p.grad.data_ptr() == buckets.slot[id(p)][1].data_ptr()
    rep = D.comm_report(steps=1)
    assert rep["world_size"] == world and rep["backend"] == "gloo"
    assert rep["grad_allreduce_calls_per_step"] == len(buckets.flats) and rep["grad_allreduce_bytes_per_step"] == nbytes
    assert rep["grad_copy_bytes_per_step"] == raw
    # a parameter outside the agreed set that receives a gradient on a later step, ON ONE RANK ONLY (a branch inactive until
    # now): no rank may raise or issue a collective on its own (the others would block in their next one).  The "late
    # gradient seen" flag rides in the last bucket; the gradient is dropped for this step where it exists (the optimiser
    # must not apply what was not exchanged), and at the next begin() BOTH ranks re-agree the set (rebuild, collective).
    for p in list(net.parameters()) + list(unused.parameters()):
        p.grad = None
    buckets.begin()
    x4 = x[:, :4]
    if rank == 1:
        (net(x).pow(2).mean() + unused(x4).pow(2).mean()).backward()
    else:
        net(x).pow(2).mean().backward()
    assert len(buckets._works) == len(buckets.flats) - 1      # the flag's bucket waits for flush()
    buckets.finish()                                          # no exception, no hang: same collectives on both ranks
    assert unused.weight.grad is None and unused.bias.grad is None
    assert buckets.late_dropped == (2 if rank == 1 else 0) and buckets.rebuilds == 0
    for p, g in zip(net.parameters(), averaged):
        assert torch.equal(p.grad, g)
    for p in list(net.parameters()) + list(unused.parameters()):
        p.grad = None
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        buckets.begin()                                       # both ranks read the averaged flag (0.5) and rebuild together
    assert buckets.rebuilds == 1 and all(id(p) in buckets.slot for p in unused.parameters())
    assert any("re-agreeing" in str(c.message) for c in caught)
    if rank == 1:
        (net(x).pow(2).mean() + unused(x4).pow(2).mean()).backward()
        lw = unused.weight.grad.clone()
    else:
        net(x).pow(2).mean().backward()
        lw = torch.zeros_like(unused.weight)
    buckets.finish()
    both = [torch.empty_like(lw) for _ in range(world)]
    dist.all_gather(both, lw)
    assert torch.allclose(unused.weight.grad, sum(both) / world, rtol=1e-6, atol=1e-8)
    for p, g in zip(net.parameters(), averaged):
        assert torch.equal(p.grad, g)
    assert unused.weight.grad.data_ptr() == buckets.slot[id(unused.weight)][1].data_ptr()
    # steady state after the rebuild: no further rebuild, the flag reads zero
    for p in list(net.parameters()) + list(unused.parameters()):
        p.grad = None
    buckets.begin()
    (net(x).pow(2).mean() + unused(x4).pow(2).mean()).backward()
    buckets.finish()
    assert buckets.rebuilds == 1
    # the replayed-backward protocol (graphs.py with N > 1): begin(launch=False) arms the hooks in copy-only mode, a HIP-graph
    # replay fills the slots without running a hook and reports its members with assume_fired(); flush() then issues every
    # bucket's collective and finish() averages — here the "replay" is a plain write into the slots
    for p in list(net.parameters()) + list(unused.parameters()):
        p.grad = None
    buckets.begin(launch=False)
    net(x).pow(2).mean().backward()                           # (a capturing iteration: hooks copy, nothing is launched)
    assert len(buckets._works) == 0
    fired = buckets.fired_ids()
    buckets.finish()
    for p, g in zip(net.parameters(), averaged):
        assert torch.equal(p.grad, g)
    buckets.begin(launch=False)
    for p in net.parameters():                                # the replay: slots rewritten in place, no hook runs
        buckets.slot[id(p)][1].fill_(float(rank + 1))
    buckets.assume_fired(fired)
    buckets.finish()
    for p in net.parameters():
        assert torch.allclose(p.grad, torch.full_like(p.grad, 1.5)), "replayed slots must be exchanged, not zeroed"
    assert unused.weight.grad is not None and float(unused.weight.grad.abs().max()) == 0.0    # member without a gradient: zeros
    for p in list(net.parameters()) + list(unused.parameters()):      # (back to the real gradients for the checks below)
        p.grad = None
    buckets.begin()
    net(x).pow(2).mean().backward()
    buckets.finish()
    # a rank-local first step: rank 1's shard leaves the second Linear without a gradient BEFORE any bucket exists —
    # the union still puts it into the buckets of both ranks
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.ReLU(), torch.nn.Linear(8, 3))
    D.broadcast_module(net2)
    b2 = D.GradBuckets(list(net2.parameters()), bucket_bytes=1 << 20)
    b2.begin()
    if rank == 0:
        net2(x).pow(2).mean().backward()
    else:
        net2[0](x).pow(2).mean().backward()
    b2.finish()
    assert all(id(p) in b2.slot for p in net2.parameters()) and net2[2].weight.grad is not None
    # SyncBN message
    xs = torch.randn(2, 5, 4, 4, generator=torch.Generator().manual_seed(7 + rank)) * (1 + rank) + rank
    C = 5
    sums = torch.cat([xs.transpose(0, 1).reshape(C, -1).sum(1), (xs.transpose(0, 1).reshape(C, -1) ** 2).sum(1)]).double()
    D.all_reduce_stats(sums)
    out[rank] = {"grads": [p.grad.clone() for p in net.parameters()], "params": [p.detach().clone() for p in net.parameters()],
                 "sums": sums, "xs": xs, "nbytes": nbytes}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_rank_data_parallel_equals_single_process():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)                          # broadcast made replicas identical
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.allclose(a, b, rtol=0, atol=0)       # averaged gradients identical on both ranks
    assert r0["nbytes"] > 0
    # single-process evaluation of the whole batch with rank 0's parameters
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    with torch.no_grad():
        for p, v in zip(net.parameters(), r0["params"]):
            p.copy_(v)
    full = make_batch(make_vocab("tiny"), BatchConfig(4, 8, 2, 4, "random"), seed=3)
    x = full[2].reshape(4, -1)[:, :6]
    net(x).pow(2).mean().backward()
    for p, g in zip(net.parameters(), r0["grads"]):
        assert torch.allclose(p.grad, g, rtol=1e-5, atol=1e-7)
    # SyncBN: reduced message + N-replica formula == the oracle's multi-replica evaluation
    import oracle
    sums = r0["sums"]
    assert torch.equal(sums, r1["sums"])
    C, n = 5, 2 * 2 * 16
    mean = sums[:C] / n
    var = (sums[C:] - sums[:C] * mean) / n
    inv_std = var.clamp(1e-5) ** -0.5
    ys = oracle.syncbn_multi_replica([r0["xs"], r1["xs"]], torch.zeros(C), torch.ones(C))
    mine0 = (r0["xs"] - mean.float().view(1, C, 1, 1)) * inv_std.float().view(1, C, 1, 1)
    assert torch.allclose(mine0, ys[0], rtol=1e-4, atol=1e-5)


def test_drain_watchdog_without_an_nccl_group_is_a_no_op():
    """dist.drain_watchdog() only has work to do under an nccl group (ProcessGroupNCCL's watchdog): with no group — or gloo —
    it reports "drained" at once and never touches the flight recorder."""
    from canonicalsg2im_amd import dist as D
    assert D.drain_watchdog(timeout_s=0.01) == "drained"
