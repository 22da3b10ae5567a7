"""HIP-graph replay of the shape-static part of the training iteration (scripts/train.py:353-393 of the reference).

Why.  One iteration is ~1 100 kernel launches; enqueueing them from Python costs ~29 ms of host time, which is the
floor of every configuration whose kernels finish sooner (BASELINE configs C2 and C4: 128x128 images or 4 images per
GPU).  Most of those launches have shapes that depend only on (batch, image size): the AttSPADE generator, the four /
five passes of the multiscale PatchGAN, their losses, their backward passes and the PatchGAN's Adam step.  They are
captured once per shape key into three HIP graphs and replayed:

    S1  generator forward (layout pyramid -> SPADE blocks -> image), PatchGAN on fake + real, GAN_Img / GAN_Feat / VGG
    S2  backward of S1's terms, with d(object-discriminator terms)/d(image) injected at the image
    S3  PatchGAN discriminator step: losses on fake.detach() + real (+ the logged "wrong" pass), backward, Adam

    S0  (round 4, late) the scene-graph encoder: forward, box-regression term, backward — one small graph per bucket of the
        batch's triplet count (rows padded to a multiple of 64 with the `__padding__` triplets every batch already carries),
        in its own memory pool; it receives no gradient from the image path because the generator consumes the
        ground-truth boxes (sg2im/meta_models.py:47 of the reference).  It takes 3.8 ms of enqueue off the host (13.4 ->
        9.6 ms per step) and nothing off the step: the host was already ahead of the device (tools/graph_timing.py)

    (round 5) S1 is captured as two graphs — the generator's forward, then the PatchGAN passes and the image terms — and the small,
    latency-bound pieces run on side streams BESIDE the replays (`CSG_GRAPH_OVERLAP=0`: one stream): the encoder (S0 or eager)
    beside S1 / S2, the object discriminator's terms for the generator beside the PatchGAN passes of S1, the generator's Adam
    step and the object discriminator's update beside S3; each is joined where its result is first needed.  C3 80.2 -> 75.9 ms,
    C4 32.8 -> 28.8 ms, C2 33.1 -> 29.3 ms per step (tools/graph_timing.py, same box); same kernels, same bits.

What stays eager, between the replays, is what follows the data: the object-crop discriminator (one crop per real object),
and the encoder on graphs of more than 2 048 triplets per sample (config C5) or in a bucket seen for the first time.  The number of
objects enters the static part only through the layout kernels' (vecs, boxes, valid) operands: they are padded to a
multiple of 32 objects with `__image__` rows, which the layout kernel culls (a culled object adds exact zeros, forward
and backward), so the results do not depend on the padding.

Nothing is skipped or reordered in a way the results can see: the spectral-norm power iterations of the PatchGAN passes
happen in the reference's order (S1: fake, real; S3: fake, real, wrong); the generator's Adam step runs eagerly after S2
over the scene-graph encoder's (eager) and the generator's (static) gradients.  Gradients of the captured parameters live
in static buffers owned by the graph set (`.grad` is re-pointed at them whenever an eager step has replaced them).

Fallback.  A batch whose key has not been captured runs `Trainer._step_eager` in the same process (a key is captured the
second time it is seen: one-off shapes never pay for a capture; the very first iteration is always eager — it creates the
optimisers' state, which a capture must find in place); `CSG_GRAPHS=0` turns the whole mechanism off.  Replay is
limited to one process per node-local GPU without an initialised process group: with N > 1 ranks the SyncBN and gradient
collectives would have to be captured by RCCL, which cannot be validated on the 1-GPU boxes — there the eager path runs.

Round 5: N > 1 ranks can replay too (backend nccl; opt-in, `CSG_GRAPHS_DIST=1` — validated on one rank only so far).  The SyncBN statistics messages are
captured where they are issued, inside S1-S3 (ProcessGroupNCCL records a collective on its own stream and joins it to the
capturing one).  The GRADIENT exchange stays outside the graphs: a capture runs under `GradBuckets.begin(launch=False)` — the
post-accumulate hooks only move gradients into their bucket slots, which the capture records as copies (most weight gradients
are written into their slots by their producers, `ops.set_grad_destinations`) — every replay reports the members it filled
(`assume_fired`), `flush()` issues all the buckets' all-reduces eagerly right after S2 (they travel under S3 and the object
discriminator's update) and after S3, and the Adam steps follow `finish()` outside the graphs.  Exercised on a 1-GPU box by a
one-rank nccl group under `CSG_DIST_FORCE=1` (every collective is issued on RCCL and is the identity; the N-replica SyncBN
formula runs): tests/test_gpu_graphs.py::test_one_rank_nccl_group_replays_with_collectives_captured.
"""
import os

import torch

from . import dist as csg_dist
from . import ops

ENABLED = os.environ.get("CSG_GRAPHS", "1") != "0"
CAPTURE_AFTER = int(os.environ.get("CSG_GRAPH_CAPTURE_AFTER", "1"))      # eager sightings of a key before it is captured
MAX_SETS = int(os.environ.get("CSG_GRAPH_MAX_SETS", "4"))
OBJ_PAD = 32
TRIPLET_PAD = int(os.environ.get("CSG_GRAPH_TRIPLET_PAD", "64"))         # the encoder's graph: triplet rows padded to a multiple of this
MAX_SG_GRAPHS = int(os.environ.get("CSG_GRAPH_MAX_SG", "8"))             # triplet-count buckets captured per shape key; 0: encoder eager
SG_MAX_TRIPLETS = int(os.environ.get("CSG_GRAPH_SG_MAX_T", "2048"))      # larger graphs (config C5: thousands of triplets per scene) are
#                                                                          GPU-bound in the encoder and rarely repeat a bucket: eager
TIMING = os.environ.get("CSG_GRAPH_TIMING") == "1"      # developer aid: per-segment host / device time of the replayed step
# the small, latency-bound pieces of the step on a second stream beside the replays (round 5): the scene-graph encoder beside
# S1 / S2, the generator's Adam step and the object discriminator's update beside S3.  0: everything on one stream
OVERLAP = os.environ.get("CSG_GRAPH_OVERLAP", "1") != "0"
SPLIT_S1 = os.environ.get("CSG_GRAPH_SPLIT_S1", "1") != "0"      # S1 as two graphs with the object terms beside the second


class _Marks:
    """Host time (perf_counter) and device time (events on the current stream) between named marks of one step."""

    def __init__(self):
        self.acc, self.steps = {}, 0
        self.cur = None

    def begin(self):
        import time
        self.cur = [("", time.perf_counter(), self._ev())]

    def _ev(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def mark(self, name):
        import time
        self.cur.append((name, time.perf_counter(), self._ev()))

    def end(self):
        torch.cuda.synchronize()
        for (_, t0, e0), (name, t1, e1) in zip(self.cur[:-1], self.cur[1:]):
            a = self.acc.setdefault(name, [0.0, 0.0])
            a[0] += 1000.0 * (t1 - t0)
            a[1] += e0.elapsed_time(e1)
        self.steps += 1

    def report(self):
        n = max(self.steps, 1)
        return {k: {"host_ms": round(v[0] / n, 3), "device_ms": round(v[1] / n, 3)} for k, v in self.acc.items()}


def _capture_mode():
    """torch.cuda.graph's capture_error_mode.  "global" (torch's default) by default: under "thread_local" — which would let
    another host thread's HIP calls (a pinned-memory loader, an async checkpoint writer) coexist with a capture — a garbage
    collection that ran INSIDE a capture aborted the process on this stack (ROCm 7.2 / torch 2.10: `Fatal Python error: Aborted
    ... Garbage-collecting` in the middle of S1, one run in three of tests/test_gpu_modules.py).  CSG_GRAPH_CAPTURE_MODE selects
    it for integrations that need it; either way the collector is paused for the duration of a capture (`_Capture`)."""
    return os.environ.get("CSG_GRAPH_CAPTURE_MODE", "global")


class _Capture:
    """`with _Capture(graph, pool): ...` = torch.cuda.graph(...) with Python's garbage collector paused: an object destroyed by
    a collection in the middle of a capture (another trainer's CUDAGraph, a tensor of a foreign pool) issues HIP calls the
    capture does not allow."""

    def __init__(self, g, pool=None):
        self.ctx = torch.cuda.graph(g, pool=pool, capture_error_mode=_capture_mode())

    def __enter__(self):
        import gc
        self.was_enabled = gc.isenabled()     # (paused, not run: a full collection costs ~100 ms of host time per capture —
        gc.disable()                          #  25 ms per step of bench.py's 4-step VGG leg, whose encoder graph is captured there)
        return self.ctx.__enter__()

    def __exit__(self, *exc):
        import gc
        try:
            return self.ctx.__exit__(*exc)
        finally:
            if self.was_enabled:
                gc.enable()


QUIESCE = {"drained": 0, "no_recorder": 0, "timeout": 0}            # how the captures of this process found the watchdog


def _quiesce_before_capture():
    """Drain the device — and, with a process group up, wait until ProcessGroupNCCL's watchdog holds no pending work: it polls
    the end events of the works on its list every ~100 ms from its own thread, and a poll that lands inside a capture killed
    the process half of the time (hipErrorCapturedEvent from WorkNCCL::isCompleted, ROCm 7.2 / torch 2.10,
    tools/probe/nccl_graph_step_probe.py).  The condition is explicit, not timed: every Work handle this package still holds
    is waited for and dropped by its owner before a capture (GradBuckets.finish(), the SyncBN handles), the device is
    synchronised, and dist.drain_watchdog() then reads the flight recorder until no entry is active — i.e. the watchdog has
    retired everything and has nothing left to query.  Only if the recorder is off (TORCH_NCCL_TRACE_BUFFER_SIZE=0 set by the
    integrator) does this fall back to a timed wait of three watchdog periods, and says so in QUIESCE.  A capture happens once
    per shape key and encoder bucket."""
    torch.cuda.synchronize()
    if csg_dist.active():
        how = csg_dist.drain_watchdog()
        QUIESCE[how] += 1
        if how != "drained":
            import time
            import warnings
            if QUIESCE[how] == 1:
                warnings.warn("graphs: ProcessGroupNCCL's flight recorder could not confirm an idle watchdog (%s); falling back "
                              "to a timed wait before HIP-graph captures" % how)
            time.sleep(float(os.environ.get("CSG_GRAPH_DRAIN_S", "0.35")))


def _pad_objects(n):
    return max(OBJ_PAD, (int(n) + OBJ_PAD - 1) // OBJ_PAD * OBJ_PAD)


def _pad_triplets(n):
    return max(TRIPLET_PAD, (int(n) + TRIPLET_PAD - 1) // TRIPLET_PAD * TRIPLET_PAD)


def _drop_stale_autograd(*roots):
    """Tensors that modules keep between iterations and that still carry the previous iteration's autograd graph: torch's
    spectral-norm hook leaves the last normalised weight on the module (`module.weight`, with its `grad_fn`), and the
    PatchGAN caches its permuted first-layer weight.  Through them the parameters' AccumulateGrad nodes of an EAGER
    iteration — bound to the stream they were created on — would be reused by the forward being captured, and the
    captured backward would have to synchronise the capture stream with that other stream.  Detaching them makes the
    capture create fresh nodes on the capture stream."""
    from torch.nn.utils.spectral_norm import SpectralNorm
    for root in roots:
        for m in root.modules():
            for hook in m._forward_pre_hooks.values():
                if isinstance(hook, SpectralNorm):
                    w = m.__dict__.get(hook.name)
                    if torch.is_tensor(w) and w.grad_fn is not None:
                        setattr(m, hook.name, w.detach())
            if getattr(m, "_w0", None) is not None:
                m._w0, m._w0_key = None, None
            m.__dict__.pop("_sn_prepared", None)


class _SgGraph:
    """S0: the scene-graph encoder's forward, the box-regression term and its backward for ONE triplet-count bucket of a
    shape key (objects padded like the rest of the set, triplet rows padded to a multiple of TRIPLET_PAD with
    `[0, __padding__, 0]` rows — the rows every batch already carries for its shorter samples: `is_edge` masks them out of
    every message, their own outputs feed nothing, their gradients are exact zeros).  Its own memory pool: it is captured
    whenever its bucket is first seen again — after S1-S3 — and replayed BEFORE them, so it must not share blocks with their
    temporaries."""

    def __init__(self, gs, tpad, pad_pred, batch):
        dev = gs.objs.device
        B = gs.objs.shape[0]
        self.gs, self.tpad = gs, tpad
        self.triplets = torch.zeros((B, tpad, 3), device=dev, dtype=batch[3].dtype)
        self.triplets[:, :, 1] = pad_pred
        self.ttype = torch.zeros((B, tpad), device=dev, dtype=batch[5].dtype)
        self.pad_pred = pad_pred
        self.graph = None
        self.grads = {}
        self.boxes_pred = self.vals = self.bbox_all = None
        self.last_used = 0
        self.fired = set()

    def load(self, triplets, triplet_type):
        T = triplets.shape[1]
        if T < self.tpad:
            self.triplets[:, T:, 0] = 0
            self.triplets[:, T:, 1] = self.pad_pred
            self.triplets[:, T:, 2] = 0
            self.ttype[:, T:] = 0
        self.triplets[:, :T].copy_(triplets, non_blocking=True)
        self.ttype[:, :T].copy_(triplet_type, non_blocking=True)

    def run(self, tr):
        gs = self.gs
        if self.graph is None:
            g = torch.cuda.CUDAGraph()
            _quiesce_before_capture()
            ops.invalidate_weight_caches()
            _drop_stale_autograd(tr.model)
            before = tr.g_buckets.fired_ids()
            with _Capture(g):
                for p in tr.sg_params:
                    p.grad = None
                self.boxes_pred = tr.model.sg_to_layout(gs.objs, self.triplets, self.ttype, gs.boxes)[1]
                out = {}
                tr.gans_model._layout_terms(out, gs.objs, gs.boxes, self.boxes_pred, None, None)
                self.bbox_all = out["bbox_pred_all"].detach()
                self.vals = out["bbox_pred"].detach()
                out["bbox_pred"].backward()
                self.boxes_pred = self.boxes_pred.detach()
            self.graph = g
            self.grads = {p: p.grad for p in tr.sg_params if p.grad is not None}
            self.fired = tr.g_buckets.fired_ids() - before          # (N > 1) the members this graph's copies fill
        else:
            tr.g_buckets.assume_fired(self.fired)
        self.graph.replay()
        for p, gr in self.grads.items():             # (another bucket's buffers, or an eager step's tensors, may be in place)
            p.grad = gr


class _GraphSet:
    """The three captured graphs of one shape key, their static inputs / outputs and gradient buffers."""

    def __init__(self, owner, key, batch):
        self.owner, self.key = owner, key
        self.graphs = {}                 # name -> torch.cuda.CUDAGraph
        self.pool = None
        self.grads = {}                  # parameter -> static gradient tensor (generator + image discriminator)
        dev = owner.tr.device
        imgs, objs, boxes = batch[0], batch[1], batch[2]
        B, O, A = objs.shape
        self.imgs = torch.empty_like(imgs)
        self.objs = torch.zeros((B, key[2], A), device=dev, dtype=objs.dtype)
        self.boxes = torch.full((B, key[2], 4), -1.0, device=dev, dtype=boxes.dtype)
        self.img = self.fake = self.d_img = self.g_vals = self.d_vals = None
        self.g_terms = self.d_terms = None
        self.sg = {}                     # padded triplet count -> _SgGraph
        self.sg_seen = {}
        self.fired = {}                  # (N > 1) graph name -> ids of the bucket members its captured backward fills
        self.bucket_generation = owner.tr.bucket_generation   # (N > 1) the flats whose slot addresses the captures bake in

    def load(self, imgs, objs, boxes):
        O = objs.shape[1]
        self.imgs.copy_(imgs, non_blocking=True)
        if O < self.objs.shape[1]:
            self.objs.zero_()                           # `__image__` rows: culled by the layout kernels
            self.boxes.fill_(-1.0)
        self.objs[:, :O].copy_(objs, non_blocking=True)
        self.boxes[:, :O].copy_(boxes, non_blocking=True)

    def run(self, name, fn):
        """Replay graph `name`; the first time, capture `fn` into it (a capture executes nothing) and then replay it, so
        the capturing iteration is an ordinary replayed iteration."""
        g = self.graphs.get(name)
        if g is None:
            g = torch.cuda.CUDAGraph()
            _quiesce_before_capture()
            ops.invalidate_weight_caches()              # every derived weight is recomputed INSIDE the graph that reads it
            _drop_stale_autograd(self.owner.tr.model, self.owner.tr.discriminator)
            with _Capture(g, self.pool):
                fn()
            if self.pool is None:
                self.pool = g.pool()
            self.graphs[name] = g
        g.replay()

    def adopt_grads(self, params):
        for p in params:
            if p.grad is not None:
                self.grads[p] = p.grad

    def point_grads(self):
        for p, g in self.grads.items():
            p.grad = g


class StepGraphs:
    """Owner of the graph sets of one `Trainer`; `step(batch)` returns the same (G, D) loss dictionaries as
    `Trainer._step_eager`, or None when the batch has to run eagerly."""

    def __init__(self, trainer):
        self.tr = trainer
        self.sets, self.seen = {}, {}
        self.active = None
        self.replays = self.eager_steps = self.captures = 0
        self.sg_replays = self.sg_captures = 0
        self.stale_drops = 0             # (N > 1) graph sets dropped because the gradient buckets were rebuilt under them
        self.side = self.side2 = None    # streams of the overlapped pieces (OVERLAP)
        self.marks = _Marks() if TIMING else None

    # ---- eligibility
    @staticmethod
    def supported(trainer):
        opt = trainer.opt
        # N > 1 ranks: replay needs collectives that can be captured (RCCL's: the SyncBN messages sit in the middle of S1-S3).
        # OPT-IN (CSG_GRAPHS_DIST=1): RCCL has executed this path on ONE rank only (tests/test_gpu_graphs.py, CSG_DIST_FORCE);
        # until it has run on a multi-GPU node the eager path — itself covered by the 2-rank tests — stays the default there
        dist_ok = not csg_dist.active() or (csg_dist.capturable() and os.environ.get("CSG_GRAPHS_DIST", "0") == "1")
        return bool(ENABLED and torch.device(trainer.device).type == "cuda" and dist_ok
                    and not opt.skip_generation and not opt.learned_converse and not (opt.mask_size or 0) > 0
                    and not getattr(opt, "freeze", 0) and hasattr(trainer.model, "layout_to_image_model"))

    def key_of(self, batch):
        imgs, objs, masks = batch[0], batch[1], batch[6]
        if masks is not None or not imgs.is_cuda:
            return None
        return (tuple(imgs.shape), int(objs.shape[0]), _pad_objects(objs.shape[1]), int(objs.shape[2]))

    def invalidate(self):
        """Drop every captured graph: needed whenever a tensor a graph reads or writes by ADDRESS is replaced — parameters
        re-created or moved (`module.to()`), optimiser state reloaded (`load_state_dict` replaces the moment tensors the
        captured Adam step updates; `Trainer.load_checkpoint` calls this), a learning rate of `optimizer_d_img` edited (it
        is baked into the captured step)."""
        self.sets, self.seen, self.active = {}, {}, None

    # ---- one iteration
    def step(self, batch):
        key = self.key_of(batch)
        gs = self.sets.get(key) if key is not None else None
        if gs is not None and gs.bucket_generation != self.tr.bucket_generation:
            # (N > 1) GradBuckets.rebuild() ran since the capture: the graphs write gradients into flats that are no longer
            # the ones flush() exchanges and p.grad points at.  Every set is stale; this iteration runs eagerly and the keys
            # are captured again on their next sightings.
            self.invalidate()
            self.stale_drops += 1
            gs = None
        if gs is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if key is None or n < CAPTURE_AFTER or len(self.sets) >= MAX_SETS or self.tr._eager_steps == 0:
                self.eager_steps += 1
                return None
            gs = _GraphSet(self, key, batch)
            self.sets[key] = gs
            self.captures += 1
            try:
                return self._run(gs, batch)
            except Exception as e:
                raise RuntimeError("HIP-graph capture of the training step failed for key %r (%s: %s); the step is half "
                                   "executed, so this is not recoverable in place — rerun with CSG_GRAPHS=0 for the eager "
                                   "path" % (key, type(e).__name__, e)) from e
        return self._run(gs, batch)

    def _object_terms_for_generator(self, img, objs, boxes):
        """GAN_Obj / GAN_Ac on the current image and d(their sum)/d(image) (the object discriminator is frozen here)."""
        leaf = img.detach().requires_grad_(True)
        terms = self.tr.gans_model.generator_object_terms(leaf, objs, boxes, None, None)
        vals = {k: v.mean() for k, v in terms.items()}
        (d_img,) = torch.autograd.grad(list(vals.values()), [leaf])
        return {k: v.detach() for k, v in vals.items()}, d_img

    def _graph_encoder(self, gs, batch, G):
        """Scene-graph encoder forward, the box-regression term and its backward: replayed from the graph of the batch's
        triplet-count bucket (S0, `_SgGraph`; a bucket is captured the second time it is seen), else eager."""
        tr = self.tr
        objs, boxes, triplets, _, triplet_type = batch[1:6]
        if not tr.model.has_graph:
            return
        sg = None
        tpad = _pad_triplets(triplets.shape[1])
        if MAX_SG_GRAPHS > 0 and tpad <= SG_MAX_TRIPLETS and "s1" in gs.graphs:      # (the set's first iteration: eager encoder)
            sg = gs.sg.get(tpad)
            if sg is None:
                seen = gs.sg_seen.get(tpad, 0)
                gs.sg_seen[tpad] = seen + 1
                if seen >= 1:
                    if len(gs.sg) >= MAX_SG_GRAPHS:                     # least recently replayed bucket makes room
                        old = min(gs.sg, key=lambda t: gs.sg[t].last_used)
                        del gs.sg[old]
                        gs.sg_seen.pop(old, None)
                    sg = gs.sg[tpad] = _SgGraph(gs, tpad, tr.model.sg_to_layout.module.vocab["pred_name_to_idx"]["__padding__"],
                                                batch)
                    self.sg_captures += 1
        if sg is not None:
            sg.load(triplets, triplet_type)
            sg.run(tr)
            sg.last_used = self.replays + self.eager_steps
            G["bbox_pred_all"] = sg.bbox_all.clone()
            G["bbox_pred"] = sg.vals.clone()
            self.boxes_pred = sg.boxes_pred[:, :objs.shape[1]]
            self.sg_replays += 1
            return
        for p in tr.sg_params:
            p.grad = None
        boxes_pred = tr.model.sg_to_layout(objs, triplets, triplet_type, boxes)[1]
        out = {}
        tr.gans_model._layout_terms(out, objs, boxes, boxes_pred, None, None)
        G["bbox_pred_all"] = out["bbox_pred_all"].detach()
        G["bbox_pred"] = out["bbox_pred"].detach()
        out["bbox_pred"].backward()
        self.boxes_pred = boxes_pred.detach()

    def _run(self, gs, batch):
        tr, opt = self.tr, self.tr.opt
        gm, gen = tr.gans_model, tr.model.layout_to_image_model
        imgs, objs, boxes = batch[0], batch[1], batch[2]
        use_obj = not opt.use_img_disc
        mk = self.marks if (self.marks is not None and len(gs.graphs) == (4 if SPLIT_S1 else 3)) else None
        if mk:
            mk.begin()
        if self.active is not gs or tr._grads_dirty:
            gs.point_grads()
            self.active, tr._grads_dirty = gs, False
        if use_obj:
            tr.discriminator.obj_discriminator.prefetch_index(objs)
        # N > 1 ranks (dist.py): the gradient exchange stays OUTSIDE the graphs — the hooks of this backward only move
        # gradients into their bucket slots (captured as copies; a replay reports the members it filled), the all-reduces are
        # issued eagerly right after S2 / S3 and travel under what follows; the SyncBN messages are captured inside S1-S3
        dp = csg_dist.active()
        if dp:
            tr.g_buckets.begin(launch=False)
        gs.load(imgs, objs, boxes)
        if mk:
            mk.mark("load+prefetch")
        G = {}
        self.boxes_pred = None
        # The encoder (E0: ~150 small dependent launches, 3 ms of device time whatever the batch) shares nothing with the
        # generator's pass — the generator consumes the ground-truth boxes (sg2im/meta_models.py:47 of the reference) — so it runs
        # on a second stream beside S1 / S2 and is joined where its loss value and its gradients are first needed (the total,
        # the generator's Adam step).  One rank only: with N > 1 its hooks and the bucket exchange stay in stream order.
        conc = OVERLAP and not dp
        main = torch.cuda.current_stream()
        if conc:
            if self.side is None:
                self.side = torch.cuda.Stream()
            self.side.wait_stream(main)                  # (the static inputs gs.load just filled)
            with torch.cuda.stream(self.side):
                self._graph_encoder(gs, batch, G)
        else:
            self._graph_encoder(gs, batch, G)
        if mk:
            mk.mark("E0 graph encoder")
        # ---- S1: generator forward + the generator's image terms (the discriminators are frozen)
        tr._d_requires_grad(False)

        # S1 is two graphs (round 5): the generator's forward, then the PatchGAN passes and the image terms.  In between the
        # object discriminator's terms for the generator (E1: crops of the fresh image through the object discriminator and
        # back to the image, 0.8 ms of small launches) start on a stream of their own and run beside the PatchGAN passes;
        # S2 — which injects their gradient at the image — waits for them.
        def s1b():
            terms = gm.generator_image_terms(gs.imgs, gs.objs, gs.boxes, None, gs.img)
            gs.g_roots = [v.mean() for v in terms.values()]
            gs.g_terms = list(terms.keys())
            gs.g_vals = torch.stack([v.detach() for v in gs.g_roots])

        def s1():
            gs.img = gen(gs.objs, gs.boxes, None, test_mode=False)
            gs.fake = gs.img.detach()
            if not SPLIT_S1:
                s1b()
        gs.run("s1", s1)
        # ---- eager: the object discriminator's terms and their gradient at the image
        obj_vals = None
        if use_obj:
            import contextlib
            if conc:
                if self.side2 is None:
                    self.side2 = torch.cuda.Stream()
                self.side2.wait_stream(main)
            with (torch.cuda.stream(self.side2) if conc else contextlib.nullcontext()):
                obj_vals, d_img = self._object_terms_for_generator(gs.img, objs, boxes)
                if gs.d_img is None:
                    gs.d_img = torch.empty_like(d_img)
                gs.d_img.copy_(d_img)

        if SPLIT_S1:
            gs.run("s1b", s1b)
        if conc and use_obj:
            main.wait_stream(self.side2)
        if mk:
            mk.mark("S1 replay + E1 object terms (G)")

        # ---- S2: backward of S1's terms (+ the injected image gradient)
        def s2():
            for p in tr.g_params:
                p.grad = None
            roots, seeds = list(gs.g_roots), [None] * len(gs.g_roots)
            if gs.d_img is not None:
                roots.append(gs.img)
                seeds.append(gs.d_img)
            torch.autograd.backward(roots, seeds)
            gs.g_roots = None
        first = "s2" not in gs.graphs
        before = tr.g_buckets.fired_ids() if dp else None
        gs.run("s2", s2)
        if first:
            gs.adopt_grads(tr.g_params)
            if dp:
                gs.fired["s2"] = tr.g_buckets.fired_ids() - before
        elif dp:
            tr.g_buckets.assume_fired(gs.fired["s2"])
        if dp:
            tr.g_buckets.flush()                    # every bucket's all-reduce goes out now and travels under S3
        if mk:
            mk.mark("S2 replay")
        # ---- the generator's Adam step (eager: one fused launch over the encoder's eager and the generator's static grads)
        if conc:
            main.wait_stream(self.side)                  # the encoder's loss value and gradients
        vals = gs.g_vals.clone()
        for i, k in enumerate(gs.g_terms):
            G[k] = vals[i]
        if obj_vals is not None:
            G.update(obj_vals)
        G["total_loss"] = torch.stack([v for k, v in G.items() if k != "bbox_pred_all"]).sum()
        if conc:
            # Adam only touches the encoder's / generator's parameters, gradients and moments; S3 reads gs.fake (detached) and
            # the discriminator: the two run side by side, and the object discriminator's update follows Adam on the side stream
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                tr.optimizer.step()
        elif not dp:
            tr.optimizer.step()
        if mk:
            mk.mark("G Adam")
        # ---- S3: the image discriminator's update
        tr._d_requires_grad(True)

        def s3():
            terms = gm.discriminator_image_terms(gs.imgs, gs.objs, gs.boxes, None, gs.fake)
            terms = {k: v.mean() for k, v in terms.items()}
            gs.d_terms = list(terms.keys())
            gs.d_vals = torch.stack([v.detach() for v in terms.values()])
            if not tr.d_frozen:
                for p in tr.d_params:
                    p.grad = None
                terms["total_img_loss"].backward()
                if not dp:                          # (N > 1: Adam follows the exchange, outside the graph)
                    tr.discriminator.optimizer_d_img.step()
        first = "s3" not in gs.graphs
        if dp and not tr.d_frozen:
            tr.d_buckets.begin(launch=False)
            before = tr.d_buckets.fired_ids()
        gs.run("s3", s3)
        if first:
            gs.adopt_grads(tr.d_params)
            if dp and not tr.d_frozen:
                gs.fired["s3"] = tr.d_buckets.fired_ids() - before
        elif dp and not tr.d_frozen:
            tr.d_buckets.assume_fired(gs.fired["s3"])
        if dp and not tr.d_frozen:
            tr.d_buckets.flush()
        if mk:
            mk.mark("S3 replay")
        vals = gs.d_vals.clone()
        D = {k: vals[i] for i, k in enumerate(gs.d_terms)}
        # ---- eager: the object discriminator's update (beside S3 on the side stream: its own networks, gs.fake and the batch)
        if use_obj:
            import contextlib
            with (torch.cuda.stream(self.side) if conc else contextlib.nullcontext()):
                terms = gm.discriminator_object_terms(imgs, objs, boxes, None, gs.fake, None)
                terms = {k: v.mean() for k, v in terms.items()}
                if not tr.d_frozen:
                    tr.discriminator.optimizer_d_obj.zero_grad(set_to_none=True)
                    tr.dobj_buckets.begin()
                    terms["total_obj_loss"].backward()
                    tr.dobj_buckets.flush()
                    if conc:
                        tr.discriminator.optimizer_d_obj.step()
                D.update({k: v.detach() for k, v in terms.items()})
        if dp and not tr.d_frozen:
            tr.d_buckets.finish()
            tr.discriminator.optimizer_d_img.step()
        if use_obj:
            if not tr.d_frozen and not conc:
                tr.dobj_buckets.finish()
                tr.discriminator.optimizer_d_obj.step()
            tr.discriminator.obj_discriminator.release_index()
        if dp:
            tr.g_buckets.finish()
            tr.optimizer.step()
        if conc:
            main.wait_stream(self.side)                  # the next step (and the caller) see the updated parameters
        if mk:
            mk.mark("E3 object D step")
            mk.end()
        # copies: gs.fake / the encoder's boxes are static buffers the next replay overwrites in place, and a caller may keep
        # last_model_out across steps (image logging every N iterations) as it may on the eager path
        tr.last_model_out = (gs.fake.clone(), None if self.boxes_pred is None else self.boxes_pred.clone(), None)
        ops.invalidate_weight_caches()              # S3's Adam step is replayed without torch's optimiser hooks
        self.replays += 1
        return G, D
