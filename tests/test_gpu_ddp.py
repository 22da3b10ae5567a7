"""Two data-parallel ranks with the REAL HIP kernels.  The GPU box has one device, so both ranks
share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU); the exchanged messages —
SyncBN (sum, sum^2) in forward and backward, bucketed gradient averaging, parameter broadcast —
and all the code around them are exactly what runs over RCCL on the 8-GPU node."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ARGV = ["--image_size", "64,64", "--ngf", "4", "--ndf", "8", "--gconv_dim", "32", "--gconv_hidden_dim", "64",
        "--gconv_num_layers", "2", "--embedding_dim", "8", "--no_vgg_loss", "--batch_size", "4", "--gpu_ids", "0,1"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(world_rank=None, single_process=False):
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    vocab = make_vocab("tiny")
    # (one process = one device id: DataParallelWithCallback refuses `--gpu_ids 0,1` without a 2-rank process group)
    opt = T.make_opt(vocab, ARGV[:-1] + ["0"] if single_process else ARGV)
    torch.manual_seed(1234 + (world_rank or 0))          # replicas are built DIFFERENT; rank 0 wins by broadcast
    tr = T.Trainer(opt, torch.device("cuda:0"))
    batch = make_batch(vocab, BatchConfig(4, 64, 2, 5, "packed"), seed=77)
    return T, tr, batch


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from canonicalsg2im_amd import dist as D
    from canonicalsg2im_amd.synth import shard_batch
    D.init_from_env(backend="gloo")
    T, tr, batch = _make(rank)
    mine = [None if t is None else t.cuda() for t in shard_batch(batch, rank, world)]
    G, Dl = tr.step(mine)
    sg, g, d = T.split_state(tr)
    out[rank] = {
        "G": {k: v.detach().float().cpu() for k, v in G.items()},
        "D": {k: v.detach().float().cpu() for k, v in Dl.items()},
        "probe": {k: g[k].detach().cpu().clone() for k in ("conv_img.weight", "up_3.conv_0.weight_orig", "fc.bias",
                                                            "up_2.norm_0.param_free_norm.running_var")},
        "dprobe": d["discriminator_0.model1.0.0.weight_orig"].detach().cpu().clone(),
        "oprobe": tr.discriminator.obj_discriminator.state_dict()["discriminator.cnn.2.weight"].detach().cpu().clone(),
    }
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_and_track_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    # replicas end the step with IDENTICAL parameters and running statistics
    for k in r0["probe"]:
        assert torch.equal(r0["probe"][k], r1["probe"][k]), k
    assert torch.equal(r0["dprobe"], r1["dprobe"]) and torch.equal(r0["oprobe"], r1["oprobe"])
    for k in ("total_loss",):
        assert torch.isfinite(r0["G"][k]).all()
    # single-process run of the whole batch from rank 0's initial weights: BatchNorm statistics are
    # global in both runs (SyncBN), so the per-rank mean losses average to the single-process losses
    # up to the N-replica clamp(var,eps) vs var+eps difference (batchnorm.py:65-68 vs :145)
    T, tr, batch = _make(0, single_process=True)
    G, Dl = tr.step([None if t is None else t.cuda() for t in batch])
    for k in ("GAN_Img", "GAN_Feat", "bbox_pred"):
        both = 0.5 * (r0["G"][k] + r1["G"][k])
        assert torch.allclose(both, G[k].detach().float().cpu(), rtol=2e-3, atol=1e-4), (k, both, G[k])
    for k in ("D_img_fake", "D_img_real"):
        both = 0.5 * (r0["D"][k] + r1["D"][k])
        assert torch.allclose(both, Dl[k].detach().float().cpu(), rtol=2e-3, atol=1e-4), (k, both, Dl[k])
    sg, g, d = T.split_state(tr)
    rv = g["up_2.norm_0.param_free_norm.running_var"].cpu()
    assert torch.allclose(r0["probe"]["up_2.norm_0.param_free_norm.running_var"], rv, rtol=1e-3, atol=1e-5)


@pytest.mark.timeout(600)
def test_bench_contract_with_two_ranks():
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per rank), with
    both ranks on cuda:0 over gloo (test hooks CSG_DIST_BACKEND / CSG_SINGLE_DEVICE): rendezvous, parameter
    broadcast, sharded steps with the SyncBN and gradient collectives, max-over-ranks timing, one JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CSG_DIST_BACKEND="gloo", CSG_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--batch", "2", "--image_size", "64", "--ngf", "8", "--ndf", "8"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["losses_finite"] and d["value"] > 0 and "cpu_baseline" not in d
    assert abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1000.0)) / d["value"] < 0.02
    # the N > 1 line can be audited from its own output: what was exchanged, by which backend, how long it blocked
    c = d["comm"]
    assert c["world_size"] == 2 and c["backend"] == "gloo"
    nb = c["grad_buckets"]
    assert nb["generator"] >= 1 and nb["d_img"] >= 1 and nb["d_obj"] >= 1
    assert c["grad_allreduce_calls_per_step"] == nb["generator"] + nb["d_img"] + nb["d_obj"]
    assert c["grad_allreduce_bytes_per_step"] > 4 * 100000         # > 100 k parameters even at ngf = ndf = 8
    # 7 SPADE resnet blocks (4 with a learned shortcut: norm_0/norm_s share their statistics) + ... : one all-reduce
    # per norm in the forward and one in the backward
    assert c["syncbn_allreduce_calls_per_step"] >= 2 * 14 and c["syncbn_allreduce_bytes_per_step"] > 0
    # weight gradients are written straight into their bucket slots (ops.set_grad_destinations): only biases, embeddings and the
    # few-output / permuted-weight layers are still copied (10.7 % of the bytes at ngf = ndf = 8, where small tensors weigh most)
    assert c["blocked_ms_per_step"] >= 0.0 and c["grad_copy_bytes_per_step"] <= 0.25 * c["grad_allreduce_bytes_per_step"]
    assert "parity_b16" not in d


@pytest.mark.timeout(600)
def test_bench_gpus_2_without_a_launcher_starts_two_ranks():
    """`python bench.py --gpus 2` with NO torchrun around it (bench.spawn_ranks): the file starts its own two ranks as a child
    torch.distributed.run before touching the GPU, relays rank 0's line and exits with the children's status.  Both ranks on
    cuda:0 over gloo (the test hooks), as above."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(CSG_DIST_BACKEND="gloo", CSG_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
           "--image_size", "64", "--ngf", "8", "--ndf", "8", "--no_vgg_variant"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["comm"]["world_size"] == 2 and d["comm"]["backend"] == "gloo" and d["losses_finite"]
    # a failing child fails the parent: an unknown config raises in every rank
    bad = subprocess.run(cmd + ["--config", "C9"], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0


# ------------------------------------------------------------------ SyncBN backward and the converse all-gather
def _syncbn_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from canonicalsg2im_amd import dist as D, ops
    D.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(500 + rank)
    C = 12
    x = (torch.randn(2, C, 6, 5, generator=g) * (1.5 + rank) + 0.3 * rank).cuda().requires_grad_(True)
    gb = (torch.randn(2, 2 * C, 6, 5, generator=g) * 0.4).cuda().requires_grad_(True)
    w = torch.randn(2, C, 6, 5, generator=g).cuda()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    y = ops.norm_act(x, gb, rm, rv, instance=False, training=True, slope=0.2, sync=True)
    (y * w).sum().backward()
    out[rank] = {k: v.detach().cpu() for k, v in dict(x=x, gb=gb, w=w, y=y, dx=x.grad, dgb=gb.grad, rm=rm, rv=rv).items()}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_syncbn_forward_and_backward_over_two_ranks_vs_oracle():
    """The SPADE norm under data parallelism (ops._NormAct with an initialised process group): forward statistics AND
    the backward's (sum dn, sum dn*xhat) are all-reduced in fp64, N-replica formula clamp(var, eps)^-1/2
    (sync_batchnorm/batchnorm.py:74-93, 128-145).  Two ranks with different shards against the oracle's multi-replica
    evaluation differentiated by autograd: outputs, dx, d(gamma||beta), running statistics."""
    import torch.nn.functional as F
    import oracle
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_syncbn_worker, args=(world, port, out), nprocs=world, join=True)
    r = [out[0], out[1]]
    C = 12
    xs = [r[i]["x"].clone().requires_grad_(True) for i in range(2)]
    gbs = [r[i]["gb"].clone().requires_grad_(True) for i in range(2)]
    rm, rv = torch.zeros(C), torch.ones(C)
    ys = oracle.syncbn_multi_replica(xs, rm, rv)
    loss = 0
    outs = []
    for i in range(2):
        yi = F.leaky_relu(ys[i] * (1 + gbs[i][:, :C]) + gbs[i][:, C:], 0.2)
        outs.append(yi)
        loss = loss + (yi * r[i]["w"]).sum()
    loss.backward()
    for i in range(2):
        assert torch.allclose(r[i]["y"], outs[i].detach(), rtol=1e-4, atol=1e-5), "y rank %d" % i
        assert torch.allclose(r[i]["dx"], xs[i].grad, rtol=1e-4, atol=1e-5), "dx rank %d" % i
        assert torch.allclose(r[i]["dgb"], gbs[i].grad, rtol=1e-4, atol=1e-5), "dgb rank %d" % i
        assert torch.allclose(r[i]["rm"], rm, rtol=1e-4, atol=1e-6) and torch.allclose(r[i]["rv"], rv, rtol=1e-4, atol=1e-6)
    assert torch.equal(r[0]["rm"], r[1]["rm"]) and torch.equal(r[0]["rv"], r[1]["rv"])


def _converse_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from canonicalsg2im_amd import dist as D
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import make_vocab
    D.init_from_env(backend="gloo")
    vocab = make_vocab("tiny")
    opt = T.make_opt(vocab, ARGV + ["--learned_converse", "1", "--skip_generation", "1"])
    torch.manual_seed(77 + rank)
    tr = T.Trainer(opt, torch.device("cuda:0"))           # broadcast makes the replicas identical
    g = torch.Generator().manual_seed(9)
    r_full = torch.rand(4, generator=g) * 3
    cc_full = torch.randint(0, 3, (4, 8, 9), generator=g).float()
    sl = slice(2 * rank, 2 * rank + 2)
    w = tr.model.sg_to_layout.module.converse_candidates_weights
    w0 = w.detach().clone()
    tr._converse_step(r_full[sl].cuda(), cc_full[sl].cuda())
    out[rank] = {"grad": w.grad.detach().cpu().clone(), "w0": w0.cpu(), "w1": w.detach().cpu().clone()}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_converse_reinforce_reward_is_normalised_over_the_global_batch():
    """--learned_converse: the per-sample reward is normalised with the mean / std of the GLOBAL batch (all-gather of
    bbox_pred_all, scripts/train.py:371-381) and the gradient averaged over ranks: equal to the single-process update
    on the whole batch."""
    from canonicalsg2im_amd.scripts.graphs_utils import calc_log_p
    from canonicalsg2im_amd.sg2im.model import get_conv_converse
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    mp.spawn(_converse_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0]["grad"], out[1]["grad"]) and torch.equal(out[0]["w1"], out[1]["w1"])
    g = torch.Generator().manual_seed(9)
    r_full = torch.rand(4, generator=g) * 3
    cc_full = torch.randint(0, 3, (4, 8, 9), generator=g).float()
    w = out[0]["w0"].clone().requires_grad_(True)
    eps = float(torch.finfo(torch.float32).eps)
    r = (r_full - r_full.mean()) / (r_full.std() + eps)
    non_meta = list(range(2, 8))
    log_prob = calc_log_p(get_conv_converse({"sg_to_layout.module.converse_candidates_weights": w}), non_meta, cc_full)
    torch.mean(r * log_prob).backward()
    assert torch.allclose(out[0]["grad"], w.grad, rtol=1e-4, atol=1e-6)
    assert not torch.equal(out[0]["w1"], out[0]["w0"])
