"""Gradient tolerance by the reference's own fp32 noise band.

Losses, images and feature maps of the path are continuous in the weights and are held to the contract's
rtol 1e-4.  Parameter GRADIENTS of the GAN objective are not: LeakyReLU / ReLU gates, the hinge and the
sign() of the L1 feature-matching term make them piecewise-constant in the activations, so the ~1e-6 forward
rounding noise of ANY fp32 implementation flips a few hundred of the 10^7..10^8 gates and moves a gradient
tensor by 1e-4..1e-2 of its norm (more the deeper the layer).  Measured on the oracle itself: the same step
evaluated in fp32 and in fp64 on the CPU differs by up to 9e-3 in relative L2 norm at ngf=64, 256x256
(profiles/archive/r02_fp64_noise_band.txt) while the losses agree to 1e-7.

The parity statement that CAN be tested is therefore: against the fp64 evaluation of the oracle, the HIP
path's gradient error is within a small factor (3x) of the fp32 reference arithmetic's own error (plus the 1e-4
contract where that error is negligible — the graph encoder's gradients, whose objective is smooth)."""
import torch


from oracle.fp64 import batch_to64, state_to64, trainstate_to64  # noqa: F401,E402  (re-exported: the tests import them from here)


def errors(x, ref64):
    """(relative L2 error, max abs error / max |ref|) of x against the fp64 reference."""
    x, r = x.detach().double().cpu(), ref64.detach().double().cpu()
    d = x - r
    return float(d.norm() / r.norm().clamp_min(1e-300)), float(d.abs().max() / r.abs().max().clamp_min(1e-300))


class Band:
    """Collects (name, hip-vs-fp64 error, fp32-reference-vs-fp64 error) rows and judges them."""

    def __init__(self, k_l2=3.0, max_cap=0.05, floor=1e-4):
        """L2 criterion: hip_l2 <= k_l2 * ref_l2 + floor (k_l2 = 3: the round-2 tables, profiles/archive/r02_band_C{3,4,5}.txt,
        hold 0 of 464 tensors beyond 3x and a worst ratio of 2.6 where the reference noise is measurable).  The max-norm is only capped (no single entry off by more
        than 5 % of the tensor's largest): one flipped gate in front of a large activation moves ONE entry of a
        weight gradient by percents of the maximum in either implementation (the fp32 oracle shows 1.4e-1 on
        G_middle_1.conv_0 against its own fp64 evaluation), so a ratio of max-norms says nothing."""
        self.k_l2, self.max_cap, self.floor = k_l2, max_cap, floor
        self.rows, self.bad = [], []

    def add(self, name, hip, ref32, ref64):
        if float(ref64.detach().abs().max()) < 1e-12:
            # analytically zero (a conv bias in front of a normalisation): rounding noise only, on every side
            assert float(hip.detach().abs().max()) < 1e-5, name + " should vanish"
            return
        h_l2, h_mx = errors(hip, ref64)
        r_l2, r_mx = errors(ref32, ref64)
        self.rows.append((name, h_l2, r_l2, h_mx, r_mx))
        if h_l2 > self.k_l2 * r_l2 + self.floor or h_mx > max(self.max_cap, 3.0 * r_mx):
            self.bad.append("%s: hip l2 %.2e (fp32 ref %.2e), hip max %.2e (fp32 ref %.2e)" % (name, h_l2, r_l2, h_mx, r_mx))

    def table(self):
        lines = ["%-64s %10s %10s %10s %10s" % ("tensor (error vs the fp64 oracle)", "hip l2", "fp32ref l2", "hip max",
                                                "fp32ref max")]
        for r in self.rows:
            lines.append("%-64s %10.2e %10.2e %10.2e %10.2e" % r)
        return "\n".join(lines)

    def check(self, tag, dump=None, outliers=2, outlier_cap=1e-2, median_cap=1.5):
        """Verdict over all tensors of one step.

        A tensor is inside the band if hip_l2 <= k_l2 * ref_l2 + floor.  The noise is made of discrete events — ONE
        flipped gate moves every gradient upstream of it (measured with tools/debug_d_c3.py: a single LeakyReLU flip in
        the 2x512x34x34 map of D0.model3 shifts the gradients of model0..model3 and of the discriminator's embedding by
        3e-4..1e-3 in relative L2 while the fp32 reference, which happened not to flip there, sits at 1e-6) — and which
        implementation draws the flip is chance.  So what is counted are EVENTS, not tensors: the tensors outside the band
        are grouped by the network they belong to (one PatchGAN scale, the object discriminator, the generator with the
        encoder behind it) — one flipped gate accounts for every outlier of its group, at most the group's five tensors in
        a discriminator scale.  At most `outliers` = TWO groups may hold outliers, with at most five tensors each (none was
        outside the band in the round-2 runs of C3, C4 and C5; profiles/archive/r05q_band_C3.txt is a run with one event on each
        side: the fp32 ORACLE at 8.4e-4 on D0.model0 — a flip behind model0 — and the HIP path at 1.4-1.8e-4 on D0.model1..3
        where the oracle drew none), as long as no tensor is off by more than 1e-2 (the fp32 reference itself reaches 9e-3
        against fp64), and the typical tensor must be as accurate as the reference's: median of hip_l2 / ref_l2 <= 1.5
        over the tensors whose reference noise is measurable (measured: 0.75 / 1.02 / 0.63).

        `outliers=None` (dense scenes, config C5: 65-129 objects per image, S = 128 layout channels, batch 2): events are
        frequent there and ONE of them moves every tensor upstream of it — the r03 run (profiles/archive/r03_band_C5.txt) has
        the fp32 reference itself at 2.4e-3 on D0.model0/1 where the HIP path is at 7.6e-4 and the HIP path at 3e-3 on all
        of D1 where the reference drew no flip.  Counting tensors says nothing in that regime; the verdict is the cap (no
        tensor beyond 1e-2) and the median ratio.  The graph encoder's rows of that scene (uniformly 2.2-3.3e-4 from fp64
        from gconvs.3 upwards, 5-28x the fp32 oracle's distance) are NOT judged here any more: tests/dev/debug_sg_c5.py
        (profiles/archive/r04_debug_sg_c5.txt) shows them to be ReLU decisions on pre-activations within rounding distance of
        zero at the input of gconvs.4 — evaluated with the HIP path's gate decisions, the fp64 oracle agrees with every
        HIP gradient to 4e-7 — and `forced_gate_rows` below holds them to that much sharper statement.  (Round 3's
        docstring blamed an event in the generator's 8x8 head; the encoder receives no gradient from the generator,
        sg2im/meta_models.py:47 of the reference.)"""
        if dump:
            import os
            os.makedirs(os.path.dirname(dump), exist_ok=True)
            with open(dump, "w") as f:
                f.write(self.table() + "\n")
        assert self.rows, tag + ": nothing compared"
        worst = max((r[1] for r in self.rows if not r[0].startswith("imgs_pred")), default=0.0)
        allowed = len(self.rows) if outliers is None else outliers
        ratios = sorted(r[1] / r[2] for r in self.rows if r[2] > 1e-5)
        median = ratios[len(ratios) // 2] if ratios else 0.0
        groups = {}
        for b in self.bad:                            # "D discriminator_0.model1.0.0.weight_orig: ..." -> "D discriminator_0"
            name = b.split(":")[0]
            groups.setdefault(name.split(".")[0] if name.startswith("D ") else name.split(" ")[0], []).append(b)
        events_ok = outliers is None or (len(groups) <= outliers and all(len(v) <= 5 for v in groups.values()))
        msg = "%s: %d of %d gradient tensors outside the fp32 noise band in %d group(s) (allowed: %s groups of at most 5), worst L2 error %.2e, median ratio %.2f\n%s" % (
            tag, len(self.bad), len(self.rows), len(groups), "any number of" if outliers is None else allowed, worst, median,
            "\n".join(self.bad[:25]))
        assert events_ok and worst <= outlier_cap and median <= median_cap, msg


# ---------------------------------------------------------------------------------------------- one training step
def grad_rows(named_params, state32, state64, skip=("repr_net", "image_encoder")):
    """[(name, hip grad, fp32 oracle grad, fp64 oracle grad)] for every parameter all three sides hold a gradient for."""
    rows = []
    for k, p in named_params:
        if any(s in k for s in skip) or state32 is None or k not in state32:
            continue
        o = state32[k]
        o64 = state64[k] if state64 is not None else None
        if p.grad is None or not torch.is_tensor(o) or o.grad is None or (o64 is not None and o64.grad is None):
            continue
        rows.append((k, p.grad, o.grad, o64.grad if o64 is not None else None))
    return rows


def step_against_oracles(tr, batch, oracle_mod, train_mod, fp64=True):
    """Run ONE `Trainer.step` on the HIP path and the same step on the oracle in fp32 and in fp64 (same weights).

    Generator / graph-encoder gradients: the three evaluations see identical inputs.  Discriminator gradients: the
    discriminator update sees the GENERATED image, |img| ~ 0.05 at initialisation, so the 1e-5 absolute fp32 noise of
    a 60-convolution generator is 1e-4..1e-3 relative and the discriminators' gradients are linear in their input.
    To test the discriminators' own arithmetic at the contract's level they are fed the same image on every side:
    the oracle's discriminator losses are evaluated (fp32 and fp64) on the image the HIP generator produced, after the
    two spectral-norm iterations of the generator-loss passes, exactly as in the step.

    `fp64=False` skips the two fp64 evaluations (minutes of CPU on config C5's S = 128 generator): rows then carry None in
    the fp64 slot and the caller judges the gradients against the fp32 oracle alone."""
    opt = tr.opt
    ts = train_mod.oracle_state_from(tr, oracle_mod)
    ts64 = trainstate_to64(ts, oracle_mod) if fp64 else None
    tsd = train_mod.oracle_state_from(tr, oracle_mod)
    tsd64 = trainstate_to64(tsd, oracle_mod) if fp64 else None
    G, D = tr.step([None if t is None else t.cuda() for t in batch])
    torch.cuda.synchronize()
    Go, Do, img_o = oracle_mod.train_step(ts, batch)
    img64 = oracle_mod.train_step(ts64, batch_to64(batch))[2] if fp64 else None
    if not opt.skip_generation:
        for state, cast in ((tsd, lambda t: t), (tsd64, lambda t: t.double())):
            if state is None:
                continue
            mo = tuple(None if t is None else cast(t.detach().cpu().float()) for t in tr.last_model_out)
            b = batch if state is tsd else batch_to64(batch)
            with torch.no_grad():
                oracle_mod.generator_losses(opt, state.d, b, mo, dobj_state=state.dobj, dmask_state=state.dmask)
            Dl = oracle_mod.discriminator_losses(opt, state.d, b, mo, dobj_state=state.dobj, dmask_state=state.dmask)
            Dl["total_img_loss"].backward()
            if state.dobj is not None:
                Dl["total_obj_loss"].backward()
            if "total_mask_loss" in Dl:
                Dl["total_mask_loss"].backward()
    g64 = (lambda st, name: getattr(st, name)) if fp64 else (lambda st, name: None)
    rows = {"SG": grad_rows(tr.model.sg_to_layout.module.named_parameters(), ts.sg, g64(ts64, "sg")) if hasattr(tr.model, "sg_to_layout") else []}
    if not opt.skip_generation:
        rows["G"] = grad_rows(tr.model.layout_to_image_model.module.named_parameters(), ts.g, g64(ts64, "g"))
        rows["D"] = grad_rows(tr.discriminator.img_discriminator.named_parameters(), tsd.d, g64(tsd64, "d"))
        if tsd.dobj is not None:
            rows["Dobj"] = grad_rows(tr.discriminator.obj_discriminator.named_parameters(), tsd.dobj, g64(tsd64, "dobj"))
        if tsd.dmask is not None:
            rows["Dmask"] = grad_rows(tr.discriminator.mask_discriminator.named_parameters(), tsd.dmask, g64(tsd64, "dmask"))
    return {"G": G, "D": D, "Go": Go, "Do": Do, "img_o": img_o, "img64": img64, "ts": ts, "ts64": ts64, "rows": rows}


def band_of(res, tr, tag, dump=None, outliers=2, **kw):
    band = Band(**kw)
    if res["img_o"] is not None:
        band.add("imgs_pred (judged by rtol 1e-4 elsewhere)", tr.last_model_out[0], res["img_o"], res["img64"])
        band.bad = []
    for group, rows in res["rows"].items():
        for k, mine, want, want64 in rows:
            band.add("%s %s" % (group, k), mine, want, want64)
    band.check(tag, dump=dump, outliers=outliers)
    return band


# ---------------------------------------------------------------------------------------------- forced ReLU gates
class GateRecorder:
    """Forward hooks on every Linear of a module tree whose ReLU is fused into its GEMM epilogue (sg2im.layers.Linear with
    `fused_slope == 0`): the sign pattern of each output, in call order — the order in which the oracle calls F.relu."""

    def __init__(self, module):
        self.gates, self.names, self.handles = [], [], []
        for name, m in module.named_modules():
            if getattr(m, "fused_slope", None) == 0.0 and hasattr(m, "weight") and m.weight.dim() == 2:
                self.handles.append(m.register_forward_hook(self._hook(name)))

    def _hook(self, name):
        def fn(mod, inp, out):
            self.gates.append((out.detach() > 0).cpu())
            self.names.append(name)
        return fn

    def remove(self):
        for h in self.handles:
            h.remove()


class forced_relu:
    """Context manager: torch.nn.functional.relu records its decisions (`seen`, `pre`) and, given `force`, replaces the
    k-th call's gate by force[k] (x * gate: same value and same derivative as a ReLU that had decided that way)."""

    def __init__(self, force=None):
        self.force, self.seen, self.pre = force, [], []

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.real = F, F.relu
        F.relu = self
        return self

    def __exit__(self, *exc):
        self.F.relu = self.real

    def __call__(self, x, inplace=False):
        k = len(self.seen)
        self.seen.append(x.detach() > 0)
        self.pre.append(x.detach())
        if self.force is not None:
            return x * self.force[k].reshape(x.shape).to(x.dtype)
        return self.real(x)


def forced_gate_rows(tr, batch, oracle_mod, train_mod, gates, state):
    """The graph encoder's box-regression gradients from the fp64 oracle evaluated with the HIP path's ReLU decisions.
    `state`: the fp32 oracle state of the encoder taken BEFORE the step.  Returns (rows [(name, hip grad, fp64 grad)],
    flip statistics [(layer, flipped units, units, max |fp64 pre-activation| at a flip / max |pre| of the layer)])."""
    import copy
    opt = copy.copy(tr.opt)
    opt.skip_generation = True
    b64 = batch_to64(batch)
    forced = state_to64(state)
    # ONE fp64 pass, with the HIP gates forced: its recorded pre-activations also give the flip statistics (a unit whose
    # forced gate disagrees with the sign of its fp64 pre-activation; upstream of it the two evaluations differ only by
    # other such units, whose pre-activations are rounding noise themselves)
    with forced_relu(force=gates) as rec:
        _, bp, _ = oracle_mod.sg2layout_forward(forced, opt.vocab, batch[1], batch[3], batch[5])
        Gl = oracle_mod.generator_losses(opt, None, b64, (None, bp, None))
    assert len(rec.seen) == len(gates), "the oracle calls F.relu %d times, the HIP encoder has %d fused ReLUs" % (
        len(rec.seen), len(gates))
    stats = []
    for name, mine, theirs, pre in zip(tr._gate_names, gates, rec.seen, rec.pre):
        diff = mine.reshape(theirs.shape) != theirs
        n = int(diff.sum())
        worst = float(pre[diff].abs().max() / pre.abs().max().clamp_min(1e-300)) if n else 0.0
        stats.append((name, n, theirs.numel(), worst))
    Gl["total_loss"].backward()
    rows = []
    for k, p in tr.model.sg_to_layout.module.named_parameters():
        if p.grad is not None and k in forced and torch.is_tensor(forced[k]) and forced[k].grad is not None:
            rows.append((k, p.grad, forced[k].grad))
    return rows, stats
