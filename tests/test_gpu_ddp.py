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


def _make(world_rank=None):
    from canonicalsg2im_amd import train as T
    from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab
    vocab = make_vocab("tiny")
    opt = T.make_opt(vocab, ARGV)
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
    T, tr, batch = _make(0)
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
