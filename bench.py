#!/usr/bin/env python3
"""Headline benchmark: 256x256 images/sec for one full G+D training step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one iteration of scripts/train.py:353-393 (graph encoder + AttSPADE generator forward,
generator losses through two PatchGAN passes, backward + Adam; discriminator losses through two more
passes, backward + Adam) on one synthetic COCO-shaped batch (config C3 of BASELINE.json: 256x256,
batch 16 per GPU, 1..30 objects per image, default widths ngf=ndf=64, the reference's default
discriminator set = image + object-crop discriminators, --no_vgg_loss because the pretrained VGG19 is
not available offline).
For N > 1 the driver launches this file under torch.distributed.run: one rank per GPU, RCCL
gradient all-reduce + SyncBN statistics, per-GPU batch fixed (weak scaling).  Rank 0 prints ONE
JSON line.  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=6)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--batch", type=int, default=16, help="images per GPU")
    p.add_argument("--image_size", type=int, default=256)
    p.add_argument("--config", default="C3", help="BASELINE config whose graph statistics to draw (C2..C5)")
    p.add_argument("--ngf", type=int, default=64)
    p.add_argument("--ndf", type=int, default=64)
    p.add_argument("--use_img_disc", type=int, default=0,
                   help="0 = the reference's default COCO/VG recipe (image + object discriminators); 1 = image only")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--vgg_loss", type=int, default=0,
                   help="1 = keep the VGG perceptual term (reference default; random-feature VGG19 here, the "
                        "pretrained weights cannot be downloaded); 0 = --no_vgg_loss, the configuration "
                        "SURVEY.md 8(d) quotes the metric on")
    p.add_argument("--no_prof", action="store_true", help="skip the per-kernel HIP-event timing")
    p.add_argument("--no_graphs", action="store_true", help="run every step eagerly (no HIP-graph replay)")
    p.add_argument("--no_vgg_variant", action="store_true", help="skip the short second measurement with the VGG loss on")
    p.add_argument("--no_c5_leg", action="store_true",
                   help="skip the short config-C5 leg (dense CLEVR graphs, batch 6) that puts the graph-encoder / layout "
                        "kernels' HBM rates (SURVEY.md 8(d), K5/K6) into the driver's line")
    p.add_argument("--no_gen_metric", action="store_true",
                   help="skip the generator-only fwd+bwd passes (use under rocprofv3 so that its per-kernel "
                        "averages cover the same launch mix as the timed region)")
    return p.parse_args()


def _physical_cores():
    """Physical cores of the host (distinct (physical id, core id) pairs of /proc/cpuinfo); os.cpu_count() if unreadable."""
    try:
        pairs, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


# The later-step parity check is taken at THIS step index of the trainer whatever --warmup / --steps say (the image grows
# away from its initial scale with every optimiser step, and with it the fp32 noise of any evaluation of it).
PARITY_LATER_STEP = 4
# Sanity bound on the ORACLE's own fp32 evaluation against its fp64 one.  fp32 noise of a 60-convolution generator measures
# 1.3e-5 .. 2.7e-5 in relative L2 (max |diff| <= 1.5e-4) over the first ten steps; a broken oracle is off by orders of
# magnitude.  These limits only say "the yardstick is an fp32 evaluation of the same function" — the verdict on the HIP
# path never depends on how close to them the oracle sits.
ORACLE_SANITY_MAX_ABS = 5e-4
ORACLE_SANITY_REL_L2 = 1e-4


def _parity(gpu_step, Go, Do, img_o, note, rel_l2_limit=2e-5, img_atol=1e-4, img64=None):
    """Losses and generated image of one GPU step against the oracle's replay of it (rtol 1e-4; image also in relative L2).
    `img64`: the oracle's image evaluated in fp64 (oracle/fp64.py) — the image is then judged against THAT at the same rule
    (two fp32 evaluations of a 60-convolution generator may each sit 0.7e-4 from the truth and 1.4e-4 from each other).
    The HIP verdict (`ok`) = losses vs the fp32 oracle + image vs its yardstick + bbox predictions; the fp32 oracle's own
    distance from fp64 is reported beside it and only bounded by the sanity limits above (`oracle_sane`)."""
    RTOL = 1e-4
    G0, D0, img0 = gpu_step
    rows, ok, max_rel, oracle_sane = {}, True, 0.0, True
    for name, mine, want in [("G." + k, G0[k], Go[k]) for k in sorted(Go) if k != "bbox_pred_all"] + \
                            [("D." + k, D0[k], Do[k]) for k in sorted(Do)]:
        a, b = float(mine), float(want.detach().mean())
        rel = abs(a - b) / max(abs(b), 1e-30)
        good = abs(a - b) <= RTOL * abs(b) + 1e-6
        rows[name] = {"hip": a, "oracle": b, "rel": float("%.3g" % rel)}
        ok, max_rel = ok and good, max(max_rel, rel if abs(b) > 1e-3 else 0.0)
    keys_ok = (set(G0) == set(Go)) and (set(D0) == set(Do))
    img_o = img_o.detach().double()

    def dist(a, ref):
        d = (a.double() - ref).abs()
        return d, float(d.norm() / ref.norm()), int((d > RTOL * ref.abs() + img_atol).sum())

    ref = img_o if img64 is None else img64.detach().double()
    d, rel_l2, over = dist(img0, ref)
    # tanh image, |img| <= 1: rtol 1e-4 plus 1e-4 of the output scale (the rule of tests/test_gpu_fullwidth.py), and —
    # a typical |pixel| at initialisation being 0.05 — relative L2 <= 2e-5 (measured 5e-6)
    img_ok = over == 0 and rel_l2 <= rel_l2_limit
    img = {"judged_against": "fp32 oracle" if img64 is None else "fp64 oracle", "max_abs_diff": float("%.3g" % d.max()),
           "rel_l2": float("%.3g" % rel_l2), "rel_l2_limit": rel_l2_limit, "atol": img_atol,
           "pixels_over_rtol_plus_atol": over, "max_abs": float("%.3g" % ref.abs().max()), "elements": int(d.numel())}
    if img64 is not None:
        d32, l2_32, over32 = dist(img_o, ref)          # the yardstick's own fp32 evaluation against fp64 ...
        dh, l2_h, overh = dist(img0, img_o)            # ... and the two fp32 evaluations against each other
        img["fp32_oracle_vs_fp64"] = {"max_abs_diff": float("%.3g" % d32.max()), "rel_l2": float("%.3g" % l2_32),
                                      "pixels_over_rtol_plus_atol": over32}
        img["hip_vs_fp32_oracle"] = {"max_abs_diff": float("%.3g" % dh.max()), "rel_l2": float("%.3g" % l2_h),
                                     "pixels_over_rtol_plus_atol": overh}
        # a broken oracle must not pass as "fp32 noise" — but fp32 noise itself must not fail a correct HIP step
        oracle_sane = float(d32.max()) <= ORACLE_SANITY_MAX_ABS and l2_32 <= ORACLE_SANITY_REL_L2
        img["fp32_oracle_vs_fp64"].update(sane=bool(oracle_sane), max_abs_limit=ORACLE_SANITY_MAX_ABS,
                                          rel_l2_limit=ORACLE_SANITY_REL_L2)
    bb = (G0["bbox_pred_all"].double() - Go["bbox_pred_all"].detach().double()).abs()
    bb_ok = bool((bb <= RTOL * Go["bbox_pred_all"].detach().double().abs() + 1e-5 * float(Go["bbox_pred_all"].abs().max())).all())
    return {"ok": bool(ok and keys_ok and img_ok and bb_ok and oracle_sane), "hip_ok": bool(ok and keys_ok and img_ok and bb_ok),
            "oracle_sane": bool(oracle_sane), "rtol": RTOL, "max_rel": float("%.3g" % max_rel),
            "losses": rows, "imgs_pred": img, "bbox_pred_all_ok": bb_ok, "note": note}


def _overlap(on):
    """The per-kernel tables time every launch with its own event pair: with the small pieces of the step on side streams
    (canonicalsg2im_amd/streams.py, graphs.py) a timed kernel would share the chip with another stream's kernels and read
    long — the table steps run everything on one stream."""
    from canonicalsg2im_amd import graphs, streams
    graphs.OVERLAP = streams.ENABLED = bool(on)


HBM_KERNELS = ("segment_avg_fwd", "segment_avg_bwd", "gather_concat_fwd", "gather_concat_bwd", "layout_fwd", "layout_bwd",
               "norm_stats", "norm_apply_fwd", "norm_bwd_reduce", "norm_bwd_dx", "act_bwd")


def hbm_table(prof_all, names=HBM_KERNELS):
    """HBM-bound kernels: algorithmic bytes (K5 messages+indices+output, K6 the layout written once, K9 the
    activation passes) over their HIP-event time."""
    hbm = {}
    for name in names:
        kms, kn, kwork = prof_all.get(name, (0.0, 0, 0.0))
        if kn and kms > 0:
            gbs = kwork / (kms * 1e-3) / 1e9
            hbm[name] = {"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4), "mb_per_launch": round(kwork / kn / 1e6, 2),
                         "us_per_launch": round(1000.0 * kms / kn, 1)}
    return hbm


def c5_leg(dev, steps=6, warmup=3, batch=6):
    """BASELINE config C5's per-GPU shard (CLEVR vocabulary, 4 attributes -> 128 layout channels, 64-128 objects per scene with
    transitive-closure graphs, 6 of the 48 images) for a few steps: the only configuration on which the graph-encoder and
    layout kernels move enough bytes for an HBM rate to mean anything."""
    import gc
    import torch
    from canonicalsg2im_amd import _lib, train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab
    base = BASELINE_CONFIGS["C5"]
    vocab, cfg = make_vocab(base["vocab"]), base["cfg"]
    opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--use_img_disc", "0", "--batch_size", str(batch),
                             "--gpu_ids", "0"])
    torch.manual_seed(0)
    tr = T.Trainer(opt, dev)
    bc = BatchConfig(batch, 256, cfg.min_objects, cfg.max_objects, cfg.graph)
    batches = [[None if t is None else t.to(dev) for t in make_batch(vocab, bc, seed=7000 + i)] for i in range(3)]
    for i in range(warmup):
        tr.step(batches[i % 3])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        G, Dl = tr.step(batches[i % 3])
    torch.cuda.synchronize()
    ms = 1000.0 * (time.perf_counter() - t0) / steps
    finite = bool(torch.isfinite(G["total_loss"]).item() and torch.isfinite(Dl["total_img_loss"]).item())
    tr.use_graphs = False
    for i in range(3):                                   # (eager warm-up: allocator and clocks in steady state before the table)
        tr.step(batches[i % 3])
    torch.cuda.synchronize()
    _overlap(False)
    _lib.prof_reset()
    _lib.prof_enable(1)
    nprof = 3
    for i in range(nprof):
        tr.step(batches[i % 3])
    torch.cuda.synchronize()
    table = _lib.prof_read()
    _lib.prof_enable(0)
    _overlap(True)
    kern = {k: {"ms_per_step": round(v[0] / nprof, 3), "launches_per_step": round(v[1] / nprof, 1)}
            for k, v in sorted(table.items(), key=lambda kv: -kv[1][0])[:12]}
    triplets = int(batches[0][3].shape[1])
    out = {"workload": "BASELINE config C5 per-GPU shard: CLEVR-shaped AttSPADE 256x256, batch %d, %d-%d objects/img, closure "
                       "graphs (%d triplet rows per image in batch 0), S=128 layout channels, default recipe, --no_vgg_loss"
                       % (batch, cfg.min_objects, cfg.max_objects, triplets),
           "ms_per_step": round(ms, 2), "value": round(batch / ms * 1000.0, 2), "unit": "img/s", "steps": steps, "warmup": warmup,
           "losses_finite": finite,
           "hbm_kernels": hbm_table(table, ("segment_avg_fwd", "segment_avg_bwd", "gather_concat_fwd", "gather_concat_bwd",
                                            "layout_fwd", "layout_bwd")),
           "top_kernels": kern,
           "note": "timed: %d steps after %d warm-up (HIP-graph replay where the shapes allow); hbm_kernels / top_kernels: %d "
                   "more eager steps with a HIP event pair on every launch; algorithmic bytes / event time against 8 TB/s"
                   % (steps, warmup, nprof)}
    del tr, batches
    gc.collect()
    torch.cuda.empty_cache()
    return out


def cpu_baseline(snapshot, batch0, gpu_step0, image_size, snapshot_k=None, gpu_stepk=None, k=None, dense=False):
    """The oracle (CPU restatement of the reference path) on the bench's own per-GPU batch, timed as the CPU baseline
    (one warm-up + three repetitions of one full G+D step) — and the checker of the benchmarked
    workload itself: the warm-up replays step 0 of the GPU trainer (`snapshot`: its weights before that step), the first
    repetition replays step k (`snapshot_k`: the weights the trainer held after k optimiser steps, taken from the live
    parameters — a stale derived-weight cache or a wrong graph replay shows here), each compared at rtol 1e-4."""
    import torch
    import oracle
    from canonicalsg2im_amd import train as T
    nimg = int(batch0[0].shape[0])
    phys = _physical_cores()

    def one(snap, threads):
        torch.set_num_threads(threads)
        ts = T.oracle_state_from(snap, oracle)
        t0 = time.time()
        out = oracle.train_step(ts, batch0)
        return time.time() - t0, out

    from oracle.fp64 import generated_image64

    def image64(snap):
        torch.set_num_threads(small)
        return generated_image64(T.oracle_state_from(snap, oracle), batch0)

    # torch's CPU convolutions stop scaling (and then regress) well before a 2 x 64-core host is full: the warm-up runs on
    # every physical core, the first repetition on 32 threads, the remaining two on whichever was faster
    small = min(phys, 32)
    t_phys, (Go, Do, img_o) = one(snapshot, phys)
    # `dense` (config C5: 128 layout channels, up to 128 overlapping objects per scene): the fp32 ORACLE is itself 1.7e-4 away
    # from fp64 on a few pixels there (profiles/archive/r04_c5_image_vs_fp64.txt; the HIP path 7e-5), so that configuration's image is
    # judged against the oracle's fp64 evaluation — same rule, rtol 1e-4 + 1e-4 absolute, 2e-5 in relative L2
    parity0 = _parity(gpu_step0, Go, Do, img_o,
                      "step 0 of the GPU trainer (taken before warm-up) vs oracle.train_step on the same weights and batch: "
                      "every loss at rtol 1e-4 (+1e-6), the whole generated image at rtol 1e-4 + 1e-4 absolute and 2e-5 in "
                      "relative L2%s" % (" against the oracle's fp64 evaluation (dense scenes)" if dense else ""),
                      img64=image64(snapshot) if dense else None)
    times, parityk = {phys: [t_phys]}, None
    if snapshot_k is not None:
        t_small, (Gk, Dk, img_k) = one(snapshot_k, small)
        times.setdefault(small, []).append(t_small)
        parityk = _parity(gpu_stepk, Gk, Dk, img_k,
                          "step %d of the same trainer (after %d optimiser steps, the captured HIP graphs replaying when they are "
                          "on) vs oracle.train_step on a snapshot of the trainer's live weights taken right before it: every loss "
                          "at rtol 1e-4 against the fp32 oracle; the image — grown away from its initial |pixel| ~ 0.05 scale by "
                          "then, tanh saturating at 1.0, so that two fp32 evaluations differ by up to 1.4e-4 on a few pixels — "
                          "against the oracle's fp64 evaluation of the same forward pass at step 0's rule (rtol 1e-4 + 1e-4 "
                          "absolute, 2e-5 in relative L2), the fp32 oracle's own distance from fp64 reported beside it" % (k, k),
                          img64=image64(snapshot_k))
        parityk["step_index"] = k
    best = min(times, key=lambda c: min(times[c]))
    reps = list(times[best]) if best != phys else []          # the all-cores run was the warm-up
    while len(reps) < 3:
        reps.append(one(snapshot, best)[0])
    reps.sort()
    dt = reps[len(reps) // 2]
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    base = {"value": round(nimg / dt, 4), "unit": "img/s", "cores": best, "kind": "port", "cpu_model": cpu_model,
            "host_cpus": os.cpu_count(), "physical_cores": phys, "repetitions": len(reps),
            "times_s": [round(t, 2) for t in reps],
            "other_thread_counts": {str(c): [round(t, 2) for t in v] for c, v in times.items() if c != best},
            "sample": "median of %d repetitions (after one warm-up on all %d physical cores) of 1 full G+D step "
                      "(oracle.train_step) on the bench's own batch 0 (%d images at %dx%d, the GPU trainer's weights), torch "
                      "CPU fp32, %d threads (the faster of {all physical cores, 32}), %.1f s per step"
                      % (len(reps), phys, nimg, image_size, image_size, best, dt)}
    return base, parity0, parityk


def spawn_ranks(n, argv):
    """`bench.py --gpus N` started WITHOUT a launcher: start N ranks of this file under torch.distributed.run as a child
    process (the reference's `--gpu_ids 0,..,N-1` contract, scripts/args.py:225-236: one command, N replicas), relay their
    output — rank 0's JSON line — and return the launcher's exit status.  Runs before anything in this process touches the
    GPU, and never replaces this process (an exec after GPU initialisation takes the box down)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    import torch
    from canonicalsg2im_amd import dist as D
    rank, world, local = D.init_from_env()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the line would not describe the run" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("CSG_SINGLE_DEVICE") == "1":        # test hook: every rank on cuda:0 (with CSG_DIST_BACKEND=gloo)
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()
    if world > 1:
        torch.distributed.barrier()
    from canonicalsg2im_amd import _lib, train as T
    from canonicalsg2im_amd.synth import BASELINE_CONFIGS, BatchConfig, make_batch, make_vocab

    base = BASELINE_CONFIGS[args.config]
    vocab = make_vocab(base["vocab"])
    cfg = base["cfg"]
    H = args.image_size
    opt_argv = ["--image_size", "%d,%d" % (H, H), "--ngf", str(args.ngf), "--ndf", str(args.ndf),
                "--use_img_disc", str(args.use_img_disc)]
    if args.vgg_loss:
        os.environ.setdefault("CSG_VGG19_RANDOM", "1")
    else:
        opt_argv.append("--no_vgg_loss")
    opt = T.make_opt(vocab, opt_argv + ["--batch_size", str(args.batch * world), "--gpu_ids",
                                        ",".join(str(i) for i in range(world))])
    torch.manual_seed(0)
    trainer = T.Trainer(opt, dev)
    bc = BatchConfig(args.batch, H, cfg.min_objects, cfg.max_objects, cfg.graph)
    nb = 4
    batches = [[None if t is None else t.to(dev) for t in make_batch(vocab, bc, seed=1000 * rank + i)]
               for i in range(nb)]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # parity of the benchmarked workload (N = 1): step 0 on batch 0 is kept (losses + generated image) together with a
    # CPU snapshot of the weights it started from; the cpu_baseline leg replays exactly that step on the oracle
    check = world == 1 and not args.no_cpu_baseline
    snapshot = batch0_cpu = gpu_step0 = None
    if check:
        snapshot = T.state_snapshot(trainer)
        batch0_cpu = make_batch(vocab, bc, seed=0)
        G0, D0 = trainer.step(batches[0])
        torch.cuda.synchronize()
        gpu_step0 = ({k: (v.detach().cpu() if k == "bbox_pred_all" else float(v.detach())) for k, v in G0.items()},
                     {k: float(v.detach()) for k, v in D0.items()}, trainer.last_model_out[0].detach().float().cpu())
        del G0, D0

    # parity after PARITY_LATER_STEP optimiser steps: a CPU snapshot of the LIVE weights, then one more (untimed) step on batch 0
    # whose losses and image the cpu_baseline leg replays on the oracle.  It is taken when the trainer has run exactly
    # PARITY_LATER_STEP steps — inside the warm-up when --warmup is long enough, after a few extra untimed steps when it is
    # not — so that neither --warmup nor --steps move the step that is checked.
    later = {"snapshot": None, "step": None, "index": None, "replayed": None}
    done = [1 if check else 0]

    def take_later_parity():
        g = trainer.graphs
        before = g.replays if g is not None else 0
        later["index"] = done[0]
        later["snapshot"] = T.state_snapshot(trainer)
        Gk, Dk = trainer.step(batches[0])
        torch.cuda.synchronize()
        later["step"] = ({k: (v.detach().cpu() if k == "bbox_pred_all" else float(v.detach())) for k, v in Gk.items()},
                         {k: float(v.detach()) for k, v in Dk.items()}, trainer.last_model_out[0].detach().float().cpu())
        later["replayed"] = bool(g is not None and g.replays > before)
        done[0] += 1

    def untimed_step(b):
        if check and later["step"] is None and done[0] == PARITY_LATER_STEP:
            take_later_parity()
        trainer.step(b)
        done[0] += 1

    graphs_on = trainer.graphs is not None and not args.no_graphs
    if trainer.graphs is not None:
        trainer.use_graphs = graphs_on
    for i in range(args.warmup):
        untimed_step(batches[i % nb])
    if graphs_on:
        # the shape-static part of the step is replayed from HIP graphs (canonicalsg2im_amd/graphs.py); a key is captured
        # the second time it is seen, so with --warmup < 2 a few more untimed steps keep the capture out of the timed region
        # (the scene-graph encoder's graph follows one sighting later: bucket of triplet counts seen once with the set in place)
        for i in range(6):
            if trainer.graphs.replays > 0 and (trainer.graphs.sg_replays > 0 or not trainer.model.has_graph):
                break
            untimed_step(batches[(args.warmup + i) % nb])
    if check:
        while later["step"] is None and done[0] < PARITY_LATER_STEP:
            untimed_step(batches[done[0] % nb])
        if later["step"] is None:
            take_later_parity()
    sync()
    snapshot_k, gpu_stepk, k_index = later["snapshot"], later["step"], later["index"]
    if not args.no_prof and not graphs_on:
        # events only around the dominant kernel inside the timed region (mode 2): a pair on every one of the
        # ~1100 launches of a step would cost ~10 ms/step of queue time and distort `value`
        _lib.prof_reset()
        _lib.prof_enable(2)
    D.comm_reset()
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        G, Dl = trainer.step(batches[i % nb])
    sync()
    elapsed = time.perf_counter() - t0
    graph_stats = eager_ms = None
    if graphs_on:
        # kernels inside a replayed graph carry no per-dispatch events: the dominant kernel's event timing (mode 2) is
        # taken over the SAME number of steps of the same batches run eagerly right after the timed region
        graph_stats = {"captures": trainer.graphs.captures, "replays": trainer.graphs.replays,
                       "eager_steps": trainer.graphs.eager_steps, "encoder_captures": trainer.graphs.sg_captures,
                       "encoder_replays": trainer.graphs.sg_replays}
        trainer.use_graphs = False
        trainer.step(batches[0])
        sync()
        if not args.no_prof:
            _lib.prof_reset()
            _lib.prof_enable(2)
        t1 = time.perf_counter()
        for i in range(args.steps):
            trainer.step(batches[i % nb])
        sync()
        eager_ms = 1000.0 * (time.perf_counter() - t1) / args.steps
    comm = D.comm_report(args.steps) if world > 1 else None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    prof, prof_all, prof_all_steps = {}, {}, 2
    if not args.no_prof:
        prof = _lib.prof_read()
        # untimed: two more steps with an event pair on EVERY launch, for the per-kernel table (one stream: _overlap)
        _overlap(False)
        _lib.prof_reset()
        _lib.prof_enable(1)
        for i in range(prof_all_steps):
            trainer.step(batches[i % nb])
        sync()
        prof_all = _lib.prof_read()
        _lib.prof_enable(0)
        _overlap(True)
    loss_ok = bool(torch.isfinite(G["total_loss"]).item() and torch.isfinite(Dl["total_img_loss"]).item())
    trainer_buckets = {"g": len(trainer.g_buckets.flats), "d": len(trainer.d_buckets.flats),
                       "dobj": len(trainer.dobj_buckets.flats) if trainer.dobj_buckets is not None else 0}

    # BASELINE.json's second metric: the SPADE generator alone, forward + backward (no optimiser step)
    gen_ms = gen_exec_ratio = None
    # (the auxiliary legs below — generator-only passes, the VGG-on variant, the C5 leg — run on ONE rank only: the N > 1 line
    # is the scaling measurement, and every extra trainer construction there is more collectives between the timed region
    # and the JSON line it has to deliver)
    if H in (64, 128, 256) and args.ngf == 64 and not args.no_gen_metric and world == 1:
        gen = trainer.model.layout_to_image_model
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            b = batches[it % nb]
            img = gen(b[1], b[2], None, test_mode=False)
            img.mean().backward()
            trainer.optimizer.zero_grad(set_to_none=True)
        e1.record()
        torch.cuda.synchronize()
        gen_ms = e0.elapsed_time(e1) / reps
        if not args.no_prof:                              # one more pass, untimed, for the executed/algorithmic FLOP ratio
            _lib.prof_reset()
            _lib.prof_enable(1)
            img = gen(batches[0][1], batches[0][2], None, test_mode=False)
            img.mean().backward()
            trainer.optimizer.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            pg = _lib.prof_read()
            _lib.prof_enable(0)
            mk = ("igemm_fwd", "igemm_fwd64", "igemm_wgrad", "gemm_nt", "gemm_tn", "wino_conv", "wino_wgrad", "wino4_conv", "wino4_wgrad")
            alg = sum(pg.get(k, (0.0, 0, 0.0))[2] for k in mk)
            exe = sum(pg.get(k, (0.0, 0, 0.0))[2] * {"wino_conv": 4.0 / 9.0, "wino_wgrad": 4.0 / 9.0, "wino4_conv": 0.25, "wino4_wgrad": 0.25}.get(k, 1.0)
                      for k in mk)
            gen_exec_ratio = exe / alg if alg > 0 else None

    # the reference's DEFAULT recipe keeps the VGG perceptual term (scripts/args.py:153-154); the headline above is
    # SURVEY.md 8(d)'s --no_vgg_loss configuration.  A short second measurement with the term on (random-feature
    # VGG19: the pretrained weights cannot be downloaded) makes that number visible to the driver as well.
    vgg_variant = None
    if not args.vgg_loss and not args.no_vgg_variant and world == 1:
        os.environ.setdefault("CSG_VGG19_RANDOM", "1")
        argv_v = [a for a in opt_argv if a != "--no_vgg_loss"]
        opt_v = T.make_opt(vocab, argv_v + ["--batch_size", str(args.batch * world), "--gpu_ids",
                                            ",".join(str(i) for i in range(world))])
        del trainer
        import gc
        gc.collect()                                      # the graph sets hold the trainer in a reference cycle
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        tv = T.Trainer(opt_v, dev)
        for i in range(2):
            tv.step(batches[i % nb])
        sync()
        t0 = time.perf_counter()
        nv = 4
        for i in range(nv):
            tv.step(batches[i % nb])
        sync()
        tvv = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(tvv, op=torch.distributed.ReduceOp.MAX)
        vgg_variant = {"ms_per_step": round(1000.0 * float(tvv.item()) / nv, 2),
                       "value": round(args.batch * world * nv / float(tvv.item()), 3), "unit": "img/s", "steps": nv,
                       "note": "same workload with the VGG perceptual loss on (reference default, scripts/args.py:153-154); "
                               "random-feature VGG19 (pretrained weights are not available offline)"}
        del tv

    c5 = None
    if world == 1 and args.config == "C3" and H == 256 and not args.no_c5_leg and not args.no_prof:
        trainer = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        c5 = c5_leg(dev)

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    imgs = args.batch * world * args.steps
    out = {
        "metric": "256x256 images/sec/node (G+D step)" if H == 256 else "%dx%d images/sec/node (G+D step)" % (H, H),
        "value": round(imgs / elapsed, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE config %s: %s-shaped AttSPADE %dx%d, batch %d/GPU, %d-%d objects/img, "
                               "ngf=%d ndf=%d, %s --use_img_disc %d; Sg2Layout GCN + SPADE G + 2-scale "
                               "PatchGAN D%s, fwd+bwd+Adam" % (args.config, base["vocab"].upper(), H, H, args.batch, cfg.min_objects,
                                                                 cfg.max_objects, args.ngf, args.ndf,
                                                                 "VGG loss (random features)" if args.vgg_loss
                                                                 else "--no_vgg_loss", args.use_img_disc,
                                                                 "" if args.use_img_disc else " + object-crop D"),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world},
        "losses_finite": loss_ok,
    }
    if graph_stats is not None:
        out["hip_graphs"] = dict(graph_stats, eager_ms_per_step=round(eager_ms, 2),
                                 note="timed region: generator + PatchGAN forward/backward/Adam replayed from 4 captured HIP "
                                      "graphs per step and the scene-graph encoder from its own (encoder_*: one per bucket of "
                                      "the batch's triplet count), object-crop discriminator enqueued eagerly "
                                      "(canonicalsg2im_amd/graphs.py); eager_ms_per_step = the same steps without replay "
                                      "(with the dominant kernel's events on)")
    # Winograd F(2x2,3x3) / F(3x3,2x2) issue 16 multiplications where the direct convolution needs 36: the kernels' `work`
    # is the ALGORITHMIC FLOP count (2*M*9*Cin*Cout, what FlopCounterMode counts for the layer); the matrix pipe EXECUTES
    # 4/9 of it.  Every `frac` below is executed FLOPs / time / peak (a hardware utilisation, <= 1 by construction);
    # the algorithmic rate is reported beside it as `algorithmic` / `algorithmic_over_peak` (it may exceed 1).
    # (F(4x4,3x3): 36 multiplications per 16 outputs where the direct sum needs 144 -> 1/4)
    # (F(3x3,4x4) weight gradient, csrc/wino4w.hip: 36 multiplications per 4x4 tile of dY where the direct sum needs 144 -> 1/4)
    EXEC = {"wino_conv": 4.0 / 9.0, "wino_wgrad": 4.0 / 9.0, "wino4_conv": 0.25, "wino4_wgrad": 0.25}
    mfma_kernels = ("igemm_fwd", "igemm_fwd64", "igemm_wgrad", "gemm_nt", "gemm_tn", "wino_conv", "wino_wgrad", "wino4_conv",
                    "wino4_wgrad")

    def mfma_rates(table, names, seconds):
        """(algorithmic TFLOP/s, executed TFLOP/s) of the launches `names` in a prof table over `seconds`."""
        alg = sum(table.get(k, (0.0, 0, 0.0))[2] for k in names)
        exe = sum(table.get(k, (0.0, 0, 0.0))[2] * EXEC.get(k, 1.0) for k in names)
        return alg / seconds / 1e12, exe / seconds / 1e12

    def roof(alg, exe, **extra):
        d = {"bound": "mfma", "achieved": round(exe, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
             "frac": round(exe / PEAK_FP32_MFMA_TFLOPS, 4), "algorithmic": round(alg, 2),
             "algorithmic_over_peak": round(alg / PEAK_FP32_MFMA_TFLOPS, 4)}
        d.update(extra)
        return d

    if prof:
        kern = {}
        for name, (ms, n, work) in prof_all.items():
            kern[name] = {"ms_per_step": round(ms / prof_all_steps, 3), "launches_per_step": round(n / prof_all_steps, 1),
                          "avg_us": round(1000.0 * ms / n, 2)}
        # the dominant kernel of the step (timed inside the timed region): the Winograd 3x3 convolution when it is on,
        # else the direct implicit GEMM
        dom = max(("wino4_conv", "wino_conv", "igemm_fwd"), key=lambda k: prof.get(k, (0.0, 0, 0.0))[0])
        ms, n, work = prof.get(dom, (0.0, 0, 0.0))
        traffic, pmc = None, {}             # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.py)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if args.config == "C3" and H == 256 and args.batch == 16 and not args.use_img_disc and not args.vgg_loss:
                # every instantiation of the dominant kernel (k_wino4_conv_v<4, true | false>: the persistent form and the
                # one-block-per-item form), weighted by the launches profiled
                key = {"wino4_conv": "k_wino4_conv_v<4", "wino_conv": "k_wino_conv2<16, 2>"}.get(dom, "k_igemm_fwd<128")
                hits = [v for k, v in pmc["kernels"].items() if k.startswith(key)]
                if sum(v["launches_profiled"] for v in hits) > 0:
                    traffic = int(round(sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in hits) /
                                        sum(v["launches_profiled"] for v in hits)))
        except (OSError, KeyError, ValueError, TypeError, ZeroDivisionError):
            pass
        if n:
            alg, exe = mfma_rates(prof, (dom,), ms * 1e-3)
            out["roofline"] = roof(
                alg, exe,
                kernel={"wino4_conv": "k_wino4_conv_v (3x3 convolutions on maps >= 32 wide, forward + backward-data, Winograd "
                                      "F(4x4,3x3) on fp32 MFMA)",
                        "wino_conv": "k_wino_conv2 (3x3 convolutions forward + backward-data, Winograd F(2x2,3x3) on fp32 "
                                     "MFMA, all instantiations)"}.get(dom, "k_igemm_fwd (conv forward + backward-data, all shapes)"),
                achieved_note="EXECUTED MFMA FLOPs (1/4 of the algorithmic 2*M*9*Cin*Cout for F(4x4,3x3), 4/9 for "
                              "F(2x2,3x3)) / kernel time of its launches (start/stop events bound to the dispatches) " +
                              ("over %d eager steps of the same batches right after the timed region (kernels inside a "
                               "replayed HIP graph carry no per-dispatch events; rocprofv3 sees them: profiles/)" % args.steps
                               if graphs_on else "inside the timed region"),
                peak_clock_note="peak = 256 CUs x 256 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md); under this load the shader "
                                "clock reads ~2.0 GHz (s_memtime vs wall clock, DESIGN.md 4.1b), where the same product is 131 TFLOP/s",
                traffic=traffic,
                traffic_source="profiles/pmc_traffic.json = " + str(pmc.get("capture", "the builder's last PMC capture")) + " (builder-run capture, not observed "
                               "by this run)" if traffic is not None else None,
                traffic_note="HBM bytes per launch (read x2-corrected FETCH_SIZE + WRITE_SIZE, two rocprofv3 --pmc passes "
                             "of this workload taken by the builder and committed as profiles/pmc_traffic.json: a constant "
                             "of that capture — a --pmc pass cannot run inside this process)",
                launches=n, avg_launch_us=round(1000.0 * ms / n, 2),
                algorithmic_gflop_per_launch=round(work / n / 1e9, 3),
                executed_gflop_per_launch=round(work * EXEC.get(dom, 1.0) / n / 1e9, 3))
        # whole step: FLOPs of every convolution / linear launch of one step over the step time
        step_alg, step_exe = mfma_rates(prof_all, mfma_kernels, prof_all_steps * elapsed / args.steps)
        if step_alg > 0:
            out["roofline_step"] = roof(
                step_alg, step_exe,
                algorithmic_tflop_per_step=round(step_alg * elapsed / args.steps, 3),
                executed_tflop_per_step=round(step_exe * elapsed / args.steps, 3),
                note="sum over every convolution / linear launch of one step (forward, backward-data, weight gradient) "
                     "/ ms_per_step")
        wnames = ("igemm_wgrad", "wino_wgrad", "wino4_wgrad")
        wms = sum(prof_all.get(k, (0.0, 0, 0.0))[0] for k in wnames)
        if wms > 0:
            walg, wexe = mfma_rates(prof_all, wnames, wms * 1e-3)
            out["roofline_wgrad"] = roof(walg, wexe, kernels="k_wino4_wgrad (F(3x3,4x4)) + k_wino_wgrad (F(3x3,2x2)) + k_igemm_wgrad",
                                         ms_per_step=round(wms / prof_all_steps, 3),
                                         winograd_ms_per_step=round(sum(prof_all.get(k, (0.0, 0, 0.0))[0] for k in
                                                                        ("wino_wgrad", "wino4_wgrad")) / prof_all_steps, 3))
        dms = sum(prof_all.get(k, (0.0, 0, 0.0))[0] for k in ("igemm_fwd", "igemm_fwd64", "igemm_wgrad"))
        if dms > 0:
            dalg, dexe = mfma_rates(prof_all, ("igemm_fwd", "igemm_fwd64", "igemm_wgrad"), dms * 1e-3)
            out["roofline_direct"] = roof(dalg, dexe, kernels="k_igemm_fwd<*> + k_igemm_wgrad<*> (4x4 PatchGAN, 1x1, small "
                                                               "3x3 convolutions, linears): forward, backward-data, weight gradient",
                                          ms_per_step=round(dms / prof_all_steps, 3))
        hbm = hbm_table(prof_all)
        out["hbm_kernels"] = hbm
        out["hbm_kernels_note"] = ("algorithmic bytes / HIP-event time; launches that move a few MB (the graph kernels on "
                                   "COCO-sized graphs: ~30 triplets per image) are launch-latency bound — their rates on "
                                   "dense graphs are in this line's `c5` object (config C5's shard, run right after)")
        out["kernels"] = kern
        out["kernels_note"] = ("per-kernel table, roofline_wgrad, roofline_direct and hbm_kernels: %d untimed steps after the "
                               "timed region with a HIP event pair on every launch (summed durations exceed the step where "
                               "the two PatchGAN scales overlap on their streams); `roofline`: events on the k_wino4_conv_v (F(4x4,3x3), F(3x3,4x4)), k_wino_conv2 and "
                               "k_igemm_fwd<128> launches only, inside the timed region" % prof_all_steps)
    if gen_ms is not None:
        # algorithmic work of SPADEGenerator fwd+bwd per image (FlopCounterMode; S = 32 or 128)
        S = len(vocab["attributes"]) * 32
        gflop = {(256, 32): 870.75, (128, 32): 217.69, (64, 32): 54.42, (256, 128): 3 * 348.37}.get((H, S))
        if gflop:
            tf = gflop * args.batch / gen_ms            # GFLOP/ms == TFLOP/s
            # executed share of those FLOPs: the launch table of one more (untimed) generator pass
            ratio = gen_exec_ratio if gen_exec_ratio else 1.0
            out["generator_fwd_bwd"] = roof(
                tf, tf * ratio, ms=round(gen_ms, 2), algorithmic_gflop_per_img=gflop, executed_over_algorithmic=round(ratio, 4),
                note="SPADEGenerator forward+backward on one %d-image batch, all kernels (convs, norms, layout, resampling) "
                     "included in the time; algorithmic FLOPs from BASELINE.md §2, executed share from the kernels' own "
                     "launch table of one such pass" % args.batch)
    if comm is not None:
        # N > 1 audit trail (rank 0's view; every rank exchanges the same messages): what travelled per step and how
        # long the compute stream sat in GradBuckets.finish() waiting for it
        comm["grad_buckets"] = {"generator": trainer_buckets["g"], "d_img": trainer_buckets["d"],
                                "d_obj": trainer_buckets["dobj"]}
        comm["note"] = ("per-step averages over the timed region on rank 0: gradient all-reduces (one per 64 MB bucket, "
                        "ReduceOp.AVG on RCCL), SyncBN fp64 (sum, sum^2) all-reduces (one per SPADE norm forward + one "
                        "per backward), converse all-gathers; grad_copy_bytes = gradients copied into their bucket "
                        "slots by the post-accumulate hooks; blocked_ms = HIP-event time inside finish() on the "
                        "compute stream (the exposed part of the exchange)")
        out["comm"] = comm
    if vgg_variant is not None:
        out["vgg_loss_variant"] = vgg_variant
    if c5 is not None:
        out["c5"] = c5
    parity_ok = True
    if check:                                             # the CPU leg runs at N = 1 only
        out["cpu_baseline"], out["parity_b16"], pk = cpu_baseline(snapshot, batch0_cpu, gpu_step0, H, snapshot_k, gpu_stepk,
                                                                    k_index, dense=cfg.graph == "closure")
        parity_ok = out["parity_b16"]["ok"]
        if pk is not None:
            pk["replayed_from_hip_graphs"] = later["replayed"]
            out["parity_step%d" % k_index] = pk
            out["parity_later_step"] = "parity_step%d" % k_index
            parity_ok = parity_ok and pk["ok"]
    print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if not parity_ok:
        raise SystemExit("bench.py: the benchmarked steps do not match the oracle (parity_b16 / parity_step*: ok = false; "
                         "hip_ok / oracle_sane say which side)")


if __name__ == "__main__":
    main()
