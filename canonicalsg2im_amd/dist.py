"""Data-parallel plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in CPU tests).

The reference's only parallel strategy is `nn.DataParallel`: parameters are
re-broadcast every forward and gradients reduce-added onto GPU 0.  Here replicas hold identical
parameters and exchange only (1) gradients — bucketed all-reduce, averaged, which equals the
reference because its per-replica losses are means over equal shards averaged with `.mean()`
(scripts/train.py:363,391) — and (2) SyncBN statistics (ops._NormAct)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default group from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # CSG_DIST_BACKEND=gloo: test hook (two ranks sharing one GPU cannot use RCCL)
            backend = os.environ.get("CSG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # ProcessGroupNCCL's flight recorder on (it is what drain_watchdog() below reads; a ring of 2000 small records)
        os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "2000")
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def active():
    """True when the data-parallel exchanges run: several ranks — or ONE rank with CSG_DIST_FORCE=1 and an initialised
    process group (bring-up and tests on a 1-GPU box: every collective is issued, on RCCL when the backend is nccl, and is the
    identity; the N-replica SyncBN formula is used)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CSG_DIST_FORCE") == "1"


def capturable():
    """Whether the collectives may be issued inside a HIP-graph capture: RCCL's are (ProcessGroupNCCL records them on its
    own stream, joined to the capturing stream); gloo's go through the host."""
    return active() and dist.get_backend() == "nccl"


def drain_watchdog(timeout_s=10.0):
    """Wait until ProcessGroupNCCL's watchdog thread holds no pending work — the condition a HIP-graph capture needs (the
    watchdog polls the end event of every work still on its list from its own thread; such a poll inside a capture aborts the
    process, graphs._quiesce_before_capture).  The device is drained first by the caller; what remains is the watchdog
    noticing, on its next 100 ms tick, that the works have completed and dropping them.  That is OBSERVABLE: the watchdog
    marks a work's flight-recorder entry `retired` in the same breath as it erases the work from its list, and
    `_dump_nccl_trace_json` lists every entry with that flag.  (NOT `onlyActive=True` / the entry's `state`: the dump itself
    queries the events to fill those in, so they turn "completed" the moment the device is idle whether or not the watchdog
    has looked — round 6 tried that first and the race stayed.)  Returns "drained" (every recorded collective is retired:
    nothing is left for the watchdog to query), "no_recorder" (no entries, or entries without the flag: the caller falls back
    to a timed wait) or "timeout"."""
    import json
    import time
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
        return "drained"
    try:
        from torch._C._distributed_c10d import _dump_nccl_trace_json as dump
    except ImportError:
        return "no_recorder"
    t0 = time.monotonic()
    while True:
        try:
            entries = json.loads(dump(includeCollectives=True, onlyActive=False)).get("entries", [])
        except (RuntimeError, ValueError, TypeError):
            return "no_recorder"
        if not entries or any("retired" not in e for e in entries):
            # (a trainer has broadcast its parameters by now: an empty recorder is a recorder that is off)
            return "no_recorder"
        if all(e["retired"] for e in entries):
            return "drained"
        if time.monotonic() - t0 > timeout_s:
            return "timeout"
        time.sleep(0.005)


# ---- communication audit (bench.py's "comm" object, tests): what was exchanged since the last comm_reset()
_COMM = {"grad_allreduce_calls": 0, "grad_allreduce_bytes": 0, "syncbn_allreduce_calls": 0, "syncbn_allreduce_bytes": 0,
         "allgather_calls": 0, "allgather_bytes": 0, "grad_copy_bytes": 0, "wait_events": []}
_AUDIT = [False]        # counters and wait events are recorded only between comm_reset() and comm_report(): a training run of
#                         a million iterations must not accumulate six timing events per step


def comm_reset():
    """Start an audit window (bench.py, tests): zero the counters and start recording."""
    for k in _COMM:
        _COMM[k] = [] if k == "wait_events" else 0
    _AUDIT[0] = True


def comm_note(kind, nbytes):
    if _AUDIT[0]:
        _COMM[kind + "_calls"] += 1
        _COMM[kind + "_bytes"] += int(nbytes)


def comm_report(steps=1):
    """Per-step averages of the counters since comm_reset(); `blocked_ms` is the HIP-event time the compute stream spent
    inside GradBuckets.finish() waiting for its collectives (the exposed, non-overlapped part of the gradient exchange)."""
    out = {"world_size": world_size(), "backend": dist.get_backend() if active() else None}
    for k, v in _COMM.items():
        if k != "wait_events":
            out[k + "_per_step"] = v / max(steps, 1)
    ms = 0.0
    for e0, e1 in _COMM["wait_events"]:
        e1.synchronize()
        ms += e0.elapsed_time(e1)
    out["blocked_ms_per_step"] = ms / max(steps, 1)
    _COMM["wait_events"] = []
    _AUDIT[0] = False                                  # the window is closed: nothing is recorded until the next comm_reset()
    return out


def all_reduce_stats(sums):
    """SyncBN exchange (reference sync_batchnorm/batchnorm.py:74-83,105-126): every replica
    contributes its per-channel (sum, sum^2) — one fp64 tensor of 2C values per norm — and gets the
    total back.  One all-reduce replaces the reference's ReduceAddCoalesced + Broadcast pair."""
    if active():
        dist.all_reduce(sums)
        comm_note("syncbn_allreduce", sums.numel() * sums.element_size())
    return sums


def all_reduce_stats_async(sums):
    """all_reduce_stats as an asynchronous collective: returns a handle whose wait() makes the current stream wait for the
    total (None on a single rank) — the caller enqueues work that does not need the statistics in between (the gamma half of
    the SPADE convolution in forward, the gamma || beta convolution's backward passes in backward), so the latency of the
    2C-value message hides under a convolution instead of stalling the compute stream."""
    if active():
        work = dist.all_reduce(sums, async_op=True)
        comm_note("syncbn_allreduce", sums.numel() * sums.element_size())
        return work
    return None


class GradBuckets:
    """Gradient exchange of one optimiser's parameters: persistent flat fp32 buckets, `.grad` kept as VIEWS into
    them, one all-reduce per bucket launched asynchronously from a post-accumulate hook as soon as the bucket's last
    gradient of the current backward has landed — the collectives of the early (deep) layers' buckets travel over
    xGMI while the rest of the backward is still computing.

    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large messages beat many small ones, so buckets are
    ~64 MB (the generator's 375 MB of gradients travel in 6 collectives).  Buckets are filled in reverse registration
    order, which is the order gradients become ready.  Which parameters take part is AGREED at the first
    synchronisation: the union over ranks of the parameters that received a gradient (one blocking all-reduce of a
    has-gradient bitmask, once) — the never-used `repr_net` / `image_encoder` of G and D stay
    out instead of tripping a DDP 'unused parameter' error, and a rank whose shard left a branch empty (no real object
    for the crop discriminator) still holds the same buckets as the others.  From then on every rank issues the same
    collectives in the same order by construction: a member without a local gradient contributes zeros and RECEIVES
    the average of the others (its `.grad` becomes the slot view like everybody's — the replicas cannot part).  A
    parameter outside the agreed set that receives a gradient later (a branch inactive at the first step) cannot be
    exchanged without the other ranks knowing, and a rank that raised on its own would leave the others blocked in their
    next collective.  So the DETECTION is collective and costs nothing: the last bucket carries one extra float, "this
    rank saw a late gradient", which travels with that bucket's all-reduce (the bucket of the first-registered parameters —
    it is held back until flush(), when the backward is over and the flag is known; its gradients are the last to be
    ready anyway).  The late gradient itself is DROPPED for this step on the ranks that hold one (`.grad = None`: the
    optimiser skips the parameter, so the replicas stay identical), and at the next begin() / flush() every rank reads
    the same averaged flag and calls `rebuild()` — a collective — together.  In steady state nothing is allocated:
    a hook copies the fresh gradient into its slot (the only extra pass over the gradients — there is no `torch.cat`,
    no copy back) and re-points `.grad` at the slot, which is what the optimiser then reads.

    Usage per backward:  begin(); loss.backward(); [independent work]; finish()  — or all_reduce_mean() alone after the
    backward (the one-shot form: the hooks were not armed, so every gradient is moved into its slot then).
    On a single rank every method returns immediately and nothing is registered."""

    def __init__(self, params, bucket_bytes=64 << 20):
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.bucket_bytes = bucket_bytes
        self.flats, self.members, self.slot = [], [], {}        # per bucket: flat tensor, [params]; param id -> (bucket, view)
        self.built = False
        self._hooks = []
        self._state = "idle"                                    # idle -> armed (begin) -> flushed (flush) -> idle (finish)
        self._pending, self._fired, self._works, self._next, self._streams = [], set(), [], 0, {}
        self.allocations = 0                                    # flat buffers ever allocated (tests: steady state adds none)
        self.rebuilds = 0
        self.generation = 0                                     # bumped whenever the flats are (re)allocated: captured HIP graphs
        #                                                         bake slot ADDRESSES in and go stale with it (graphs.py)
        self.late_dropped = 0                                   # late gradients dropped (one step each) before a rebuild
        self._launch_in_hooks = True
        self._flag = self._flag_host = self._flag_event = None  # the "late gradient seen" float riding in the last bucket
        self._flag_pending = False
        self._late_ids = set()                                  # parameters whose late gradient this rank dropped

    # ---- construction (first synchronisation, or a parameter's first gradient)
    def _build(self):
        for h in self._hooks:
            h.remove()
        self.generation += 1
        wanted = set(self.slot) | self._late_ids                # a rebuild keeps the members and adds the late arrivals
        self._late_ids = set()
        self._hooks, self.flats, self.members, self.slot = [], [], [], {}
        # the live set is the UNION over ranks (blocking, but only here): every rank builds identical buckets even if
        # its own shard left a branch without a gradient
        has = torch.tensor([1 if (p.grad is not None or id(p) in wanted) else 0 for p in self.params], dtype=torch.int32,
                           device=self.params[0].device if (self.params and dist.get_backend() == "nccl") else "cpu")
        if has.numel():
            dist.all_reduce(has, op=dist.ReduceOp.MAX)
        live = [p for p, h in zip(self.params, has.tolist()) if h]
        # parameters that are pieces of ONE allocation (SPADE's gamma / beta convolution weights: one joined tensor, one
        # weight-gradient launch) stay adjacent, in memory order, so that the joined gradient has one contiguous slot
        by_storage = {}
        for p in live:
            by_storage.setdefault(p.untyped_storage().data_ptr(), []).append(p)
        order, seen = [], set()
        for p in reversed(live):                                 # reverse registration order ~ order of readiness
            if id(p) in seen:
                continue
            fam = sorted(by_storage[p.untyped_storage().data_ptr()], key=lambda q: q.data_ptr())
            for q in fam:
                seen.add(id(q))
                order.append(q)
        cur, size, groups = [], 0, []
        for p in order:
            n = p.numel() * 4
            if cur and size + n > self.bucket_bytes:
                groups.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += n
        if cur:
            groups.append(cur)
        al = lambda n: (n + 3) // 4 * 4                          # every slot starts on a 16-byte boundary (kernels write there)
        self._offs = {}
        for b, group in enumerate(groups):
            extra = 4 if b == len(groups) - 1 else 0             # the late-gradient flag (+ padding to 16 bytes)
            flat = torch.zeros(sum(al(p.numel()) for p in group) + extra, device=group[0].device, dtype=torch.float32)
            self.allocations += 1
            off = 0
            for p in group:
                self._offs[id(p)] = off
                view = flat[off:off + p.numel()]
                # the slot mirrors the parameter's own (dense) memory layout — conv weights are channels-last — so that
                # the fused optimiser walks parameter, gradient and moments in the same element order
                view = view.view(p.shape) if p.is_contiguous() else view.as_strided(p.shape, p.stride())
                off += al(p.numel())
                self.slot[id(p)] = (b, view)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))
            self.flats.append(flat)
            self.members.append(group)
        # where the weight-gradient producers write (ops.set_grad_destinations): each parameter's slot, and for runs of
        # parameters that are adjacent in memory AND in their bucket the joined slot as well
        self._dests = {}
        for b, group in enumerate(self.members):
            run = None                                           # [first data_ptr, numel so far, flat offset]
            for p in group:
                n, off = p.numel(), self._offs[id(p)]
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if dense:
                    self._dests[(p.data_ptr(), n)] = self.flats[b][off:off + n]
                    if run is not None and run[0] + run[1] * 4 == p.data_ptr() and run[2] + run[1] == off:
                        run[1] += n
                        self._dests[(run[0], run[1])] = self.flats[b][run[2]:run[2] + run[1]]
                    else:
                        run = [p.data_ptr(), n, off]
                else:
                    run = None
        self._flag = self.flats[-1][-4:-3] if self.flats else None
        self._flag_pending = False
        if self._flag is not None and self._flag.is_cuda:
            self._flag_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._flag_event = torch.cuda.Event()
        else:
            self._flag_host = self._flag_event = None
        self.built = True

    def _resolve_flag(self):
        """COLLECTIVE when it fires: the flag of the previous exchange says some rank saw a gradient outside the agreed
        set — every rank reads the same value at the same point (the start of its next backward) and rebuilds."""
        if not self._flag_pending:
            return
        self._flag_pending = False
        if self._flag_event is not None:
            self._flag_event.synchronize()                       # (recorded a whole step ago)
            seen = float(self._flag_host[0]) > 0.0
        else:
            seen = float(self._flag_host) > 0.0
        if seen:
            import warnings
            warnings.warn("GradBuckets: a parameter outside the agreed set received a gradient on some rank during the last "
                          "step (it was dropped for that step on every rank); re-agreeing the set now (rebuild #%d)"
                          % (self.rebuilds + 1))
            self.rebuild()

    def resolve(self):
        """COLLECTIVE when it fires (see _resolve_flag): read the late-gradient flag of the previous exchange NOW.  Trainer.step
        calls it for every bucket set at the top of an iteration, whichever path (eager or graph replay) the iteration then
        takes: every rank rebuilds at the same point, and a replay can compare `generation` before it touches a graph."""
        if active() and self.built:
            self._resolve_flag()
        return self.generation

    def _launch_ready(self, force=False):
        """Launch, in bucket order (identical on every rank), the all-reduces of the buckets that are complete."""
        while self._next < len(self.flats) and (force or self._pending[self._next] == 0):
            b = self._next
            if b == len(self.flats) - 1 and not force:           # carries the late-gradient flag: known at flush() only
                break
            if self._pending[b]:                                 # parameters that got no gradient this time contribute 0
                for p in self.members[b]:
                    if id(p) not in self._fired:
                        self.slot[id(p)][1].zero_()
            if self.flats[b].is_cuda:
                # gradients may have been written on other streams (the PatchGAN's half-resolution scale runs — and is
                # back-propagated — on a side stream): the collective is enqueued behind everything those streams hold
                cur = torch.cuda.current_stream(self.flats[b].device)
                for st in self._streams.get(b, ()):
                    if st != cur:
                        cur.wait_stream(st)
            self._works.append(dist.all_reduce(self.flats[b], op=_avg_op(), async_op=True))
            comm_note("grad_allreduce", self.flats[b].numel() * 4)
            self._next += 1

    def _hook(self, p):
        if self._state != "armed":
            return
        b, view = self.slot[id(p)]
        if p.grad is not view:
            if p.grad.data_ptr() != view.data_ptr() or p.grad.stride() != view.stride():
                view.copy_(p.grad)                               # (producers without a destination: biases, embeddings, ...)
                if _AUDIT[0]:
                    _COMM["grad_copy_bytes"] += p.grad.numel() * 4
            p.grad = view
        if view.is_cuda:
            self._streams.setdefault(b, set()).add(torch.cuda.current_stream(view.device))
        if id(p) not in self._fired:
            self._fired.add(id(p))
            self._pending[b] -= 1
        if self._launch_in_hooks:
            self._launch_ready()

    def _arm(self):
        self._pending = [len(g) for g in self.members]
        self._fired, self._works, self._next = set(), [], 0
        self._streams = {}
        self._state = "armed"
        if self.flats and self.flats[0].is_cuda:
            from . import ops                                    # the producers of this backward write into the slots
            ops.set_grad_destinations(self._dests)

    # ---- per-backward protocol
    def begin(self, launch=True):
        """Call right before `backward()` (after zero_grad): arms the hooks for this backward.  `launch=False`: the hooks
        only move gradients into their slots and launch nothing — the form a HIP-graph capture records (graphs.py): the
        copies become graph nodes, a replay reports its members with assume_fired(), and flush() issues every collective
        eagerly after the replay."""
        if not active() or not self.built:
            return
        self._resolve_flag()
        self._arm()
        self._launch_in_hooks = launch

    def assume_fired(self, ids):
        """A HIP-graph REPLAY ran part of this backward: no hook fired, but the captured producers and copies filled the slots
        of the members `ids` (the set fired_ids() returned after the capturing iteration) — they count as fired."""
        if not active() or self._state != "armed":
            return
        for pid in ids:
            if pid in self.slot and pid not in self._fired:
                self._fired.add(pid)
                self._pending[self.slot[pid][0]] -= 1

    def fired_ids(self):
        return set(self._fired)

    def flush(self):
        """After `backward()`: launch whatever has not been launched yet; the collectives keep running."""
        if not active() or self._state == "flushed":
            return
        if not self.built or self._state == "idle":
            # first synchronisation, or the one-shot form (begin() was not called for this backward, so the hooks did
            # nothing): every gradient is an ordinary tensor, or a slot view autograd accumulated into
            if not self.built:
                self._build()
            else:
                self._resolve_flag()
            self._arm()
            for group in self.members:
                for p in group:
                    if p.grad is not None:
                        self._hook(p)
        late = [p for p in self.params if p.grad is not None and id(p) not in self.slot]
        if late and self._flag is None:
            # no bucket exists to carry the flag (nothing had a gradient when the set was agreed): nothing has been issued
            self._state = "flushed"
            raise RuntimeError(
                "GradBuckets: %d parameter(s) received their first gradient after an EMPTY set was agreed (shapes %s); call "
                "rebuild() on EVERY rank (a collective)" % (len(late), [tuple(p.shape) for p in late[:4]]))
        if self._flag is not None:
            self._flag.fill_(1.0 if late else 0.0)
        for p in late:                                           # not exchanged this step -> not applied on any rank
            p.grad = None
            self._late_ids.add(id(p))
        self.late_dropped += len(late)
        self._launch_ready(force=True)
        self._state = "flushed"

    def finish(self):
        """Wait for the collectives; gradients are then the mean over ranks — on EVERY member, also those this rank's
        backward left without a gradient.  Returns the bytes exchanged."""
        if not active():
            return 0
        self.flush()
        timed = _AUDIT[0] and bool(self.flats) and self.flats[0].is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if timed:
            e1.record()
            _COMM["wait_events"].append((e0, e1))
        if not _has_avg():
            n = world_size()
            for flat in self.flats:
                flat.div_(n)
        if self._flag is not None:                               # read at the start of the next backward (_resolve_flag)
            if self._flag_event is not None:
                self._flag_host.copy_(self._flag, non_blocking=True)
                self._flag_event.record()
            else:
                self._flag_host = self._flag.clone()
            self._flag_pending = True
        for group in self.members:
            for p in group:
                if id(p) not in self._fired:             # no local gradient: the others' average is this rank's gradient too
                    p.grad = self.slot[id(p)][1]
        self._works = []
        self._state = "idle"
        self._launch_in_hooks = True
        if self.flats and self.flats[0].is_cuda:
            from . import ops
            ops.clear_grad_destinations()
        return sum(f.numel() * 4 for f in self.flats)

    def rebuild(self):
        """COLLECTIVE: re-agree the live set (every rank must call it at the same point, between two steps) — after
        flush() reported a parameter outside the set, or when a branch is switched on.  Gradients currently held are kept."""
        if not active():
            return
        held = {id(p): p.grad.clone() for p in self.params if p.grad is not None}
        self._build()
        for q in self.params:
            if id(q) in held and id(q) in self.slot:
                view = self.slot[id(q)][1]
                view.copy_(held[id(q)])
                q.grad = view
        self._state = "idle"
        self.rebuilds += 1

    def all_reduce_mean(self):
        """grad <- mean over ranks.  With begin() before the backward the buckets were filled (and partly sent) by the
        hooks; without it this is the one-shot exchange of whatever gradients the parameters hold now."""
        return self.finish()


def _has_avg():
    return dist.get_backend() == "nccl"            # RCCL implements ncclAvg; gloo has no averaging reduction


def _avg_op():
    return dist.ReduceOp.AVG if _has_avg() else dist.ReduceOp.SUM


def broadcast_module(module, src=0):
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    from . import ops                                  # in-place edit through .data: derived weight caches are stale now
    ops.invalidate_weight_caches()
