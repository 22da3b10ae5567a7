"""Why do the graph-encoder gradients of the C5 whole-step test (tests/test_gpu_fullwidth.py::
test_c5_full_generator_step_vs_oracle, profiles/archive/r03q_band_C5.txt) sit 5-28x further from fp64 than the fp32 oracle,
uniformly from gconvs.3 upwards, while gconvs.4.net2 / box_net are clean?

The encoder's only objective is the box regression (the generator consumes the ground-truth boxes), so the scene of that
test is replayed through the encoder alone: HIP modules vs the oracle in fp32 and fp64, with every ReLU decision of the
four Linear layers of every GraphTripleConv recorded on all three sides.  Then the fp64 oracle is evaluated once more with
its gates FORCED to the HIP path's decisions: if the gradient error collapses, the excess is gate flips (discrete events
on pre-activations within rounding distance of zero), not an accumulation problem of the row-sum kernels.

    python tests/dev/debug_sg_c5.py            (on the GPU box)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
import oracle  # noqa: E402
from canonicalsg2im_amd import train as T  # noqa: E402
from canonicalsg2im_amd.synth import BatchConfig, make_batch, make_vocab  # noqa: E402
from fp64_band import errors, state_to64  # noqa: E402

seed, bseed = int(os.environ.get("SEED", "6")), int(os.environ.get("BSEED", "8"))
vocab = make_vocab("clevr")
opt = T.make_opt(vocab, ["--image_size", "256,256", "--no_vgg_loss", "--use_img_disc", "1", "--batch_size", "2"])
torch.manual_seed(seed)
tr = T.Trainer(opt, torch.device("cuda:0"))
ts = T.oracle_state_from(tr, oracle)
s32, s64 = ts.sg, state_to64(ts.sg)
batch = make_batch(vocab, BatchConfig(2, 256, 64, 128, "closure"), seed=bseed)
imgs, objs, boxes, triplets, _, ttype = batch[:6]
print("objects per image:", (objs[..., 0] != 0).sum(1).tolist(), "triplets:", tuple(triplets.shape))

# ---- HIP side: every Linear of the gconv MLPs and of box_net reports its post-activation output
model = tr.model.sg_to_layout.module
hip_masks, names = [], []
for li, layer in enumerate(model.gconvs):
    for net in ("net1", "net2"):
        for idx in (0, 2):
            lin = getattr(layer, net)[idx]
            names.append("gconvs.%d.%s.%d" % (li, net, idx))
            lin.register_forward_hook(lambda m, i, o, dst=hip_masks: dst.append((o.detach() > 0).cpu()))
names.append("box_net.0")
model.box_net[0].register_forward_hook(lambda m, i, o, dst=hip_masks: dst.append((o.detach() > 0).cpu()))
dev = lambda t: t.cuda()
obj_vecs, boxes_pred, _ = model(dev(objs), dev(triplets), dev(ttype), dev(boxes))
out = {}
tr.gans_model._layout_terms(out, dev(objs), dev(boxes), boxes_pred, None, None)
out["bbox_pred"].backward()
print("ReLU layers recorded on the HIP side:", len(hip_masks))


# ---- oracle side: F.relu wrapped to record (and optionally force) the gate decisions in call order
class Gates:
    def __init__(self, force=None):
        self.force, self.seen, self.pre = force, [], []
        self.real = F.relu

    def __call__(self, x, inplace=False):
        i = len(self.seen)
        self.seen.append((x.detach() > 0))
        self.pre.append(x.detach())
        if self.force is not None:
            return x * self.force[i].reshape(x.shape).to(x.dtype)
        return self.real(x)


def bbox_loss(opt, objs, boxes, boxes_pred):
    """Pix2PixModel._layout_terms restated for CPU tensors of either precision (sg2im/pix2pix_model.py:71-85)."""
    per = F.smooth_l1_loss(boxes_pred.reshape(-1, 4), boxes.reshape(-1, 4).to(boxes_pred.dtype), reduction='none')
    ids = objs.reshape(-1, objs.size(-1))
    real = ((ids.sum(1, keepdim=True) != 0) if ids.size(1) > 1 else (ids != 0)).to(per.dtype)
    per = per * opt.bbox_pred_loss_weight * real
    B, O = boxes.shape[0], boxes.shape[1]
    return (per.view(B, O, 4).sum(dim=[1, 2]) / real.view(B, O).sum(dim=1)).mean()


def run(state, force=None):
    g = Gates(force)
    F.relu = g
    try:
        ov, bp, _ = oracle.sg2layout_forward(state, vocab, objs, triplets, ttype)
        loss = bbox_loss(opt, objs, boxes, bp)
        loss.backward()
    finally:
        F.relu = g.real
    return g, bp, loss


g32, bp32, l32 = run(s32)
g64, bp64, l64 = run(s64)
print("loss hip %.9g fp32 %.9g fp64 %.12g" % (float(out["bbox_pred"]), float(l32), float(l64)))
assert len(g64.seen) == len(hip_masks), (len(g64.seen), len(hip_masks))
print("\nReLU decisions that differ from the fp64 oracle (units; |fp64 pre-activation| of the differing units):")
for name, mh, m32, m64, pre in zip(names, hip_masks, g32.seen, g64.seen, g64.pre):
    mh = mh.reshape(m64.shape)
    dh, d3 = (mh != m64), (m32 != m64)
    if int(dh.sum()) or int(d3.sum()):
        mag = pre[dh].abs()
        print("  %-18s hip %4d  fp32 %4d  of %9d   max|pre| at hip flips %.2e (layer max %.2e)"
              % (name, int(dh.sum()), int(d3.sum()), m64.numel(), float(mag.max()) if mag.numel() else 0.0, float(pre.abs().max())))


def table(title, ref_state):
    rows = []
    for k, p in model.named_parameters():
        if p.grad is None or k not in ref_state or ref_state[k].grad is None:
            continue
        rows.append((k, errors(p.grad, ref_state[k].grad)[0], errors(s32[k].grad, ref_state[k].grad)[0]))
    print("\n" + title)
    for k, eh, e3 in rows:
        if "gconvs.0" in k or "gconvs.2.net2" in k or "gconvs.3" in k or "gconvs.4" in k or "box_net" in k or "emb" in k:
            print("  %-52s hip %.2e   fp32 oracle %.2e" % (k, eh, e3))
    worst = max(r[1] for r in rows)
    print("  worst hip error %.2e" % worst)
    return worst


w_free = table("relative L2 error of the gradients against the fp64 oracle (its own gates):", s64)
# fp64 oracle with the HIP path's gate decisions
s64f = state_to64(ts.sg)
run(s64f, force=[m for m in hip_masks])
w_forced = table("against the fp64 oracle evaluated with the HIP path's gate decisions:", s64f)
print("\nverdict: worst error %.2e -> %.2e when the fp64 oracle takes the HIP gates" % (w_free, w_forced))
