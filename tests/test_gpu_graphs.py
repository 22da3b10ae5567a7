"""HIP-graph replay of the shape-static part of the step (canonicalsg2im_amd/graphs.py) against the eager path and the
oracle: same losses, same parameters after several optimiser steps, eager fallback on a new shape in the same process."""
import copy
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cuda():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _make(cuda, argv, graphs):
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import make_vocab
    vocab = make_vocab("tiny")
    opt = T.make_opt(vocab, ["--image_size", "64,64", "--ngf", "8", "--ndf", "8", "--gconv_dim", "32",
                             "--gconv_hidden_dim", "64", "--gconv_num_layers", "2", "--embedding_dim", "8",
                             "--no_vgg_loss", "--batch_size", "4"] + argv)
    torch.manual_seed(0)
    tr = T.Trainer(opt, cuda)
    if not graphs:
        tr.graphs = None
    return vocab, tr


def _batches(vocab, cuda, n, B=4, lo=2, hi=6, size=64):
    from canonicalsg2im_amd.synth import BatchConfig, make_batch
    return [[None if t is None else t.to(cuda) for t in make_batch(vocab, BatchConfig(B, size, lo, hi, "packed"), seed=10 + i)]
            for i in range(n)]


def _same_weights(src, dst):
    dst.model.load_state_dict(copy.deepcopy(src.model.state_dict()))
    dst.discriminator.load_state_dict(copy.deepcopy(src.discriminator.state_dict()))
    from canonicalsg2im_amd import ops
    ops.invalidate_weight_caches()


@pytest.mark.parametrize("use_img_disc", [0, 1])
def test_graph_replay_matches_eager(cuda, use_img_disc):
    """Six iterations (eager, capture, four replays) on three alternating batches against six eager iterations from the
    same weights.  Image-discriminator-only recipe: the first captured iteration reproduces the eager one to 1e-6; later
    iterations (and the default recipe from its second iteration on) within the Adam sign-noise tolerance of
    tests/test_gpu_modules.py (STEP2).  tests/test_graph_step_vs_oracle pins the captured iteration to the oracle."""
    argv = ["--use_img_disc", str(use_img_disc)]
    vocab, eager = _make(cuda, argv, graphs=False)
    _, graphed = _make(cuda, argv, graphs=True)
    assert graphed.graphs is not None, "graph replay should be available on one GPU with the default objective"
    _same_weights(eager, graphed)
    bs = _batches(vocab, cuda, 3)
    for it in range(6):
        b = bs[it % 3]
        Ge, De = eager.step(b)
        Gg, Dg = graphed.step(b)
        assert list(Ge.keys()) == list(Gg.keys()), (list(Ge.keys()), list(Gg.keys()))
        assert list(De.keys()) == list(Dg.keys()), (list(De.keys()), list(Dg.keys()))
        # (with the object discriminator the crop backward scatters with float atomics — the only atomics on the path —
        # so two EAGER runs already part at the first Adam step: sign noise on near-zero gradients)
        tol = 1e-6 if it <= (1 if use_img_disc else 0) else 2e-3
        if not use_img_disc and it > 2:
            continue                 # the two trajectories part like two eager runs do (sign noise compounds per Adam step):
            #                          per-iteration correctness of the replayed path is test_graph_step_vs_oracle's job
        for name, e, g in [("G." + k, Ge[k], Gg[k]) for k in Ge] + [("D." + k, De[k], Dg[k]) for k in De]:
            e, g = e.detach().float().cpu(), g.detach().float().cpu()
            assert e.shape == g.shape, name
            assert torch.allclose(g, e, rtol=tol, atol=tol * 1e-1), \
                "iteration %d %s: graph %s vs eager %s" % (it, name, g.flatten()[:4].tolist(), e.flatten()[:4].tolist())
        ie, ig = eager.last_model_out[0], graphed.last_model_out[0]
        assert torch.allclose(ig, ie, rtol=tol, atol=tol), "iteration %d image: max abs diff %g" % (it, float((ig - ie).abs().max()))
    assert graphed.graphs.captures == 1 and graphed.graphs.replays == 5 and graphed.graphs.eager_steps == 1
    # parameters after six Adam steps: elements whose true gradient is zero move by +-lr per step on rounding noise
    lr = 1e-4
    for (n, pe), (_, pg) in zip(list(eager.model.named_parameters()) + list(eager.discriminator.named_parameters()),
                                list(graphed.model.named_parameters()) + list(graphed.discriminator.named_parameters())):
        d = (pe.detach() - pg.detach()).abs().max().item()
        assert d <= 2.2 * 6 * max(lr, 1e-2 if "candidates_weights" in n else lr), "%s differs by %g" % (n, d)


def test_graph_step_vs_oracle(cuda):
    """The captured iteration itself (S1/S2/S3 replays + the eager pieces) against the CPU oracle's step on the weights
    the trainer held before it: every loss at rtol 1e-4."""
    import oracle
    from canonicalsg2im_amd import train as T
    vocab, tr = _make(cuda, ["--use_img_disc", "0"], graphs=True)
    bs = _batches(vocab, cuda, 2)
    tr.step(bs[0])                                   # eager (first sighting of the key)
    for it in range(2):                              # capture, then a replay — each checked against the oracle
        ts = T.oracle_state_from(tr, oracle)
        G, D = tr.step(bs[(it + 1) % 2])
        cpu_batch = [None if t is None else t.cpu() for t in bs[(it + 1) % 2]]
        Go, Do, img_o = oracle.train_step(ts, cpu_batch)
        for k in Go:
            if k == "bbox_pred_all":
                continue
            a, b = float(G[k]), float(Go[k].detach().mean())
            assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, "G %s: graph %g vs oracle %g (iteration %d)" % (k, a, b, it)
        for k in Do:
            a, b = float(D[k]), float(Do[k].detach().mean())
            assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, "D %s: graph %g vs oracle %g (iteration %d)" % (k, a, b, it)
        d = (tr.last_model_out[0].cpu().double() - img_o.detach().double())
        assert float(d.norm() / img_o.detach().double().norm()) <= 2e-5
    assert tr.graphs.replays == 2


def test_new_shape_runs_eagerly_in_process(cuda):
    """A batch with another batch size (a new key) runs eagerly, the captured key keeps replaying afterwards, and the
    gradients the optimisers read are the right buffers on both sides of the switch."""
    vocab, tr = _make(cuda, ["--use_img_disc", "0"], graphs=True)
    a = _batches(vocab, cuda, 2, B=4)
    b = _batches(vocab, cuda, 1, B=2)
    c = _batches(vocab, cuda, 1, B=4, lo=33, hi=40)          # more than 32 objects: another padded width, another key
    for batch in (a[0], a[1], a[0], b[0], a[1], c[0], a[0]):
        G, D = tr.step(batch)
        assert torch.isfinite(G["total_loss"]).item() and torch.isfinite(D["total_img_loss"]).item()
    g = tr.graphs
    assert g.captures == 1 and g.replays == 4 and g.eager_steps == 3, (g.captures, g.replays, g.eager_steps)


def test_graphs_can_be_switched_off(cuda, monkeypatch):
    from canonicalsg2im_amd import graphs
    monkeypatch.setattr(graphs, "ENABLED", False)
    _, tr = _make(cuda, ["--use_img_disc", "0"], graphs=True)
    assert tr.graphs is None


def test_checkpoint_load_drops_the_captured_graphs(cuda):
    """`optimizer.load_state_dict` replaces the moment tensors a captured Adam step updates in place: loading a checkpoint
    must drop the graphs (they are captured again on the next repeated shape), and training continues from the loaded
    state exactly as an eager trainer does.  (Image-discriminator recipe: no float atomics on the path, so the two
    trainers can be compared at every iteration.)"""
    vocab, a = _make(cuda, ["--use_img_disc", "1"], graphs=True)
    _, b = _make(cuda, ["--use_img_disc", "1"], graphs=False)
    _same_weights(a, b)
    bs = _batches(vocab, cuda, 2)
    for it in range(3):
        a.step(bs[it % 2])
    for it in range(5):                               # the checkpoint comes from a DIFFERENT history than a's own
        b.step(bs[(it + 1) % 2])
    assert a.graphs.replays == 2
    ck = copy.deepcopy(b.checkpoint_dict(t=5))        # the eager trainer's state, into the graphed trainer ...
    for key in ("optim_state", "d_img_optim_state"):  # ... in the form the reference's plain Adam writes it
        for g in ck[key]["param_groups"]:
            g.update(fused=None, capturable=False, foreach=None)
        for st in ck[key]["state"].values():
            st["step"] = st["step"].detach().cpu()
    a.load_checkpoint(ck)
    assert not a.graphs.sets
    g0 = a.discriminator.optimizer_d_img.param_groups[0]
    assert g0["fused"] and g0["capturable"], "the loaded param_groups must not switch the captured step's flags off"
    for it in range(3):                               # eager (first sighting after the drop), capture, replay
        Ga, Da = a.step(bs[it % 2])
        Gb, Db = b.step(bs[it % 2])
        # the first two reproduce the eager trainer; the third has one replayed optimiser step behind it, whose rounding
        # differs from the eager one's in the last bit and parts the two through Adam's sign noise (as in
        # test_graph_replay_matches_eager) — 2e-2 there; what a stale graph would do is checked on the moments below
        tol = 1e-5 if it < 2 else 2e-2
        for name, x, y in [("G." + k, Ga[k], Gb[k]) for k in Gb] + [("D." + k, Da[k], Db[k]) for k in Db]:
            assert torch.allclose(x.float().cpu(), y.float().cpu(), rtol=tol, atol=tol * 1e-1), (it, name, x, y)
    assert a.graphs.captures == 2
    # the moments the re-captured Adam step updates are the LOADED ones: same step count and first moments as the eager
    # trainer's (a graph kept across the load would have left them at their checkpoint values)
    oa, ob = a.discriminator.optimizer_d_img, b.discriminator.optimizer_d_img
    checked = 0
    for pa, pb in zip(oa.param_groups[0]["params"], ob.param_groups[0]["params"]):
        sa, sb = oa.state[pa], ob.state[pb]
        assert ("step" in sa) == ("step" in sb)
        if "step" not in sb:                          # a parameter no loss reaches: Adam keeps no state for it
            continue
        checked += 1
        assert float(sa["step"]) == float(sb["step"]) == 8.0
        ma, mb = sa["exp_avg"].float().cpu(), sb["exp_avg"].float().cpu()
        assert float((ma - mb).norm()) <= 5e-2 * float(mb.norm()) + 1e-7, float((ma - mb).norm() / mb.norm())
    assert checked >= 10


def test_encoder_graph_matches_the_eager_encoder(cuda, monkeypatch):
    """S0: the scene-graph encoder replayed from its own graph (triplet rows padded to the bucket with `__padding__`
    triplets) against the same iterations with the encoder enqueued eagerly between the replays: box-regression losses to
    1e-6 at every iteration, the encoder's parameters after six optimiser steps within Adam's sign noise."""
    from canonicalsg2im_amd import graphs
    argv = ["--use_img_disc", "1"]
    vocab, a = _make(cuda, argv, graphs=True)
    _, b = _make(cuda, argv, graphs=True)
    _same_weights(a, b)
    bs = _batches(vocab, cuda, 2)
    out = []
    for tr, max_sg in ((a, 8), (b, 0)):
        monkeypatch.setattr(graphs, "MAX_SG_GRAPHS", max_sg)
        rows = []
        for it in range(6):
            G, _ = tr.step(bs[it % 2])
            rows.append((G["bbox_pred"].detach().float().cpu(), G["bbox_pred_all"].detach().float().cpu()))
        out.append(rows)
    assert a.graphs.sg_captures >= 1 and a.graphs.sg_replays >= 2 and b.graphs.sg_replays == 0
    for it, ((la, alla), (lb, allb)) in enumerate(zip(*out)):
        tol = 1e-6 if it < 4 else 2e-3                  # (after the first replayed encoder step the two runs are two fp32 trajectories)
        assert torch.allclose(la, lb, rtol=tol, atol=tol * 1e-1), (it, la, lb)
        assert torch.allclose(alla, allb, rtol=tol, atol=tol * 1e-1), (it, alla, allb)
    lr = 1e-4
    for (n, pa), (_, pb) in zip(a.model.sg_to_layout.named_parameters(), b.model.sg_to_layout.named_parameters()):
        d = (pa.detach() - pb.detach()).abs().max().item()
        assert d <= 2.2 * 6 * max(lr, 1e-2 if "candidates_weights" in n else lr), "%s differs by %g" % (n, d)


def _one_rank_nccl_worker(rank, port, out):
    """A ONE-rank nccl (RCCL) process group with CSG_DIST_FORCE=1: every data-parallel exchange of the N > 1 path is issued —
    bucketed ReduceOp.AVG gradient all-reduces, the fp64 SyncBN statistics messages (N-replica formula), the late-gradient
    flag — and is the identity, so the replayed trainer must track the eager one exactly as it does without a group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      CSG_DIST_FORCE="1", CSG_GRAPHS_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      TORCH_NCCL_TRACE_BUFFER_SIZE="2000", TORCH_FR_BUFFER_SIZE="2000")   # (dist.init_from_env sets these)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    from canonicalsg2im_amd import dist as D
    assert D.active() and D.capturable() and D.world_size() == 1
    cuda = torch.device("cuda:0")
    res = {}
    for use_img_disc in (0, 1):
        argv = ["--use_img_disc", str(use_img_disc)]
        vocab, eager = _make(cuda, argv, graphs=False)
        _, graphed = _make(cuda, argv, graphs=True)
        assert graphed.graphs is not None, "replay must be available with an nccl group up"
        _same_weights(eager, graphed)
        bs = _batches(vocab, cuda, 2)
        rows = []
        D.comm_reset()
        for it in range(5):
            Ge, De = eager.step(bs[it % 2])
            Gg, Dg = graphed.step(bs[it % 2])
            if os.environ.get("CSG_TEST_TRACE"):
                print("recipe", use_img_disc, "iteration", it, "done", flush=True)
            rows.append(({k: float(v) for k, v in Ge.items() if v.numel() == 1}, {k: float(v) for k, v in Gg.items() if v.numel() == 1},
                         {k: float(v) for k, v in De.items()}, {k: float(v) for k, v in Dg.items()}))
        torch.cuda.synchronize()
        rep = D.comm_report(steps=10)                      # (both trainers' steps)
        pe = torch.cat([p.detach().flatten() for p in eager.model.parameters()] +
                       [p.detach().flatten() for p in eager.discriminator.img_discriminator.parameters()])
        pg = torch.cat([p.detach().flatten() for p in graphed.model.parameters()] +
                       [p.detach().flatten() for p in graphed.discriminator.img_discriminator.parameters()])
        res[use_img_disc] = {"rows": rows, "captures": graphed.graphs.captures, "replays": graphed.graphs.replays,
                             "comm": rep, "param_rel": float((pe - pg).norm() / pe.norm()),
                             "g_buckets": len(graphed.g_buckets.flats), "rebuilds": graphed.g_buckets.rebuilds,
                             "quiesce": dict(__import__("canonicalsg2im_amd.graphs", fromlist=["QUIESCE"]).QUIESCE),
                             "grads_are_slots": all(p.grad is None or p.grad.data_ptr() == graphed.g_buckets.slot[id(p)][1].data_ptr()
                                                    for p in graphed.g_buckets.params if id(p) in graphed.g_buckets.slot)}
        if use_img_disc == 0:
            # a bucket REBUILD between two replays (what the late-gradient protocol does from begin() / resolve()): the flats are
            # re-allocated, the captured graphs still address the old ones.  The next step must notice (generation counter),
            # drop the sets, run eagerly, and capture again — and the gradients Adam reads must be the exchanged ones.
            old_ptrs = [f.data_ptr() for f in graphed.g_buckets.flats]
            keep = list(graphed.g_buckets.flats)             # (held: the allocator must not hand the same addresses back)
            graphed.g_buckets.rebuild()
            moved = all(f.data_ptr() not in old_ptrs for f in graphed.g_buckets.flats)
            rows2 = []
            for it in range(5, 10):
                Ge, De = eager.step(bs[it % 2])
                Gg, Dg = graphed.step(bs[it % 2])
                rows2.append(({k: float(v) for k, v in Ge.items() if v.numel() == 1},
                              {k: float(v) for k, v in Gg.items() if v.numel() == 1}))
            torch.cuda.synchronize()
            pe = torch.cat([p.detach().flatten() for p in eager.model.parameters()])
            pg = torch.cat([p.detach().flatten() for p in graphed.model.parameters()])
            gb = graphed.g_buckets
            res["rebuild"] = {"moved": moved, "stale_drops": graphed.graphs.stale_drops, "captures": graphed.graphs.captures,
                              "replays": graphed.graphs.replays, "rows": rows2, "param_rel": float((pe - pg).norm() / pe.norm()),
                              "generation": gb.generation,
                              "grads_are_slots": all(p.grad is None or p.grad.data_ptr() == gb.slot[id(p)][1].data_ptr()
                                                     for p in gb.params if id(p) in gb.slot)}
            del keep
    out.update(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_one_rank_nccl_group_replays_with_collectives_captured():
    """N > 1 readiness on a 1-GPU box (graphs.py, round 5): with a process group up the step is still replayed from HIP graphs;
    the SyncBN all-reduces are captured inside them (RCCL), the gradient all-reduces (ReduceOp.AVG on RCCL) are issued eagerly
    around the replays.  One rank, so every collective is the identity: the replayed trainer matches the eager one."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_one_rank_nccl_worker, args=(port, out), nprocs=1, join=True)
    for use_img_disc in (0, 1):
        r = out[use_img_disc]
        assert r["captures"] == 1 and r["replays"] == 4, (r["captures"], r["replays"])
        assert r["rebuilds"] == 0 and r["g_buckets"] >= 1 and r["grads_are_slots"]
        # every capture found the watchdog idle by READING it (flight recorder), not by sleeping
        q = r["quiesce"]
        assert q["drained"] >= 4 and q["no_recorder"] == 0 and q["timeout"] == 0, q
        c = r["comm"]
        assert c["backend"] == "nccl" and c["grad_allreduce_calls_per_step"] >= 2 and c["syncbn_allreduce_calls_per_step"] > 10, c
        for it, (Ge, Gg, De, Dg) in enumerate(r["rows"]):
            tol = 1e-6 if it <= 1 else 2e-2                   # (later iterations: the Adam sign-noise band of the other tests)
            for k in Ge:
                assert abs(Ge[k] - Gg[k]) <= tol * abs(Ge[k]) + 1e-5, (use_img_disc, it, k, Ge[k], Gg[k])
            for k in De:
                assert abs(De[k] - Dg[k]) <= tol * abs(De[k]) + 1e-5, (use_img_disc, it, k, De[k], Dg[k])
        assert r["param_rel"] < 5e-3, r["param_rel"]
    # captured graphs must not outlive the flats they address
    rb = out["rebuild"]
    assert rb["moved"] and rb["generation"] == 2 and rb["stale_drops"] == 1, rb
    assert rb["captures"] == 2 and rb["replays"] >= 4 + 2, (rb["captures"], rb["replays"])     # dropped, re-captured, replayed again
    assert rb["grads_are_slots"]
    for it, (Ge, Gg) in enumerate(rb["rows"]):
        for k in Ge:
            assert abs(Ge[k] - Gg[k]) <= 2e-2 * abs(Ge[k]) + 1e-5, ("after rebuild", it, k, Ge[k], Gg[k])
    assert rb["param_rel"] < 5e-3, rb["param_rel"]
