#!/usr/bin/env python3
"""Generate the committed golden vectors by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference); nothing of the
reference travels: the outputs are `.npz` files of plain tensors + JSON
metadata.  Usage:  python tests/golden/make_golden.py

The stub shim below is the one recorded in SURVEY.md appendix A: it satisfies
imports of packages that are absent here (torchvision, cv2, h5py, ...) and that
the hot path never executes.
"""
import json
import os
import sys
import types

import numpy as np
import torch
import PIL.Image  # noqa: F401  (sg2im/data/utils.py needs the submodule imported)

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Any:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return a[0] if a else None

    def __getattr__(self, n):
        return _Any


_mod("torch.tensor", Tensor=torch.Tensor)
tv = _mod("torchvision")
tv.transforms = _mod("torchvision.transforms", Normalize=_Any, Compose=_Any, Resize=_Any, ToTensor=_Any)
tv.models = _mod("torchvision.models", vgg19=_Any)
_mod("torchvision.models.inception", inception_v3=_Any)
_mod("torchvision.utils", save_image=_Any)
for _n in ("cv2", "h5py", "pycocotools", "pycocotools.mask"):
    _mod(_n)
_mod("tensorboardX", SummaryWriter=_Any)
_mod("imageio", imwrite=_Any, imread=_Any)

import warnings  # noqa: E402
warnings.filterwarnings("ignore")

from scripts.args import parser as ref_parser, init_args as ref_init_args  # noqa: E402  (reference)
from sg2im.graph import GraphTripleConv, get_predicates_weights  # noqa: E402
from sg2im.layout import boxes_to_layout, masks_to_layout  # noqa: E402
from sg2im.model import Sg2LayoutModel  # noqa: E402
from sg2im.pix2pix_model import Pix2PixModel  # noqa: E402
from spade.models.networks.architecture import SPADEResnetBlock  # noqa: E402
from spade.models.networks.discriminator import AcCropDiscriminator, MultiscaleDiscriminator  # noqa: E402
from spade.models.networks.generator import SPADEGenerator  # noqa: E402
from spade.models.networks.normalization import SPADE  # noqa: E402
from spade.models.networks.sync_batchnorm import SynchronizedBatchNorm2d  # noqa: E402

from canonicalsg2im_amd.synth import BatchConfig, deterministic_state, make_batch, make_vocab  # noqa: E402  (ours: inputs only)


def ref_opt(vocab, argv):
    args = ref_parser.parse_args(argv)
    args.vocab = vocab
    args.gpu_ids = "-1"
    ref_init_args(args)
    return args


def npy(t):
    return t.detach().cpu().clone().numpy()


def sd_np(module, prefix="", skip=()):
    return {prefix + k: npy(v) for k, v in module.state_dict().items() if not any(s in k for s in skip)}


def grads_np(module, prefix="grad:", skip=()):
    seen, out = {}, {}
    for k, p in module.named_parameters():
        if any(s in k for s in skip) or p.grad is None:
            continue
        out[prefix + k] = npy(p.grad)
    return out


def shapes_of(module, skip=()):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()
            if not any(u in k for u in skip)}


def save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrays)
    print("%-28s %8.1f KB  %d arrays" % (name + ".npz", os.path.getsize(path) / 1024, len(arrays)))


# ----------------------------------------------------------------------------- fixtures
def fx_layout():
    torch.manual_seed(0)
    vecs = torch.randn(6, 8, requires_grad=True)
    boxes = torch.tensor([[0.10, 0.20, 0.50, 0.40],
                          [0.00, 0.00, 1.00, 1.00],
                          [0.60, 0.55, 0.35, 0.30],
                          [0.30, 0.30, 0.07, 0.90],     # thin, runs off the bottom edge
                          [-0.2, 0.40, 0.60, 0.20],     # starts left of the image
                          [0.45, 0.05, 0.50, 0.11]], requires_grad=True)
    arrays = {"vecs": npy(vecs), "boxes": npy(boxes)}
    for H, W in ((16, 16), (32, 32), (24, 40)):
        out = boxes_to_layout(vecs, boxes, H, W)
        w = torch.randn_like(out)
        gv, gb = torch.autograd.grad((out * w).sum(), [vecs, boxes])
        tag = "%dx%d" % (H, W)
        arrays.update({"out_" + tag: npy(out), "w_" + tag: npy(w), "gvecs_" + tag: npy(gv), "gboxes_" + tag: npy(gb)})
    save("layout", {"sizes": [[16, 16], [32, 32], [24, 40]], "ref": "sg2im/layout.py:12-45"}, **arrays)


def fx_masks_layout():
    torch.manual_seed(6)
    vecs = torch.randn(5, 8, requires_grad=True)
    boxes = torch.tensor([[0.10, 0.20, 0.50, 0.40], [0.00, 0.00, 1.00, 1.00], [0.60, 0.55, 0.35, 0.30],
                          [0.30, 0.30, 0.07, 0.90], [-0.2, 0.40, 0.60, 0.20]])
    masks = (torch.rand(5, 16, 16) > 0.4).long()                    # collate gives int64 masks
    soft = torch.rand(5, 16, 16)                                    # mask-net style float masks
    arrays = {"vecs": npy(vecs), "boxes": npy(boxes), "masks": npy(masks), "soft": npy(soft)}
    for H in (32, 64):
        for tag, m in (("int", masks), ("soft", soft)):
            out = masks_to_layout(vecs, boxes, m, H, H)
            w = torch.randn_like(out)
            (gv,) = torch.autograd.grad((out * w).sum(), [vecs])
            arrays.update({"out_%s_%d" % (tag, H): npy(out), "w_%s_%d" % (tag, H): npy(w), "gvecs_%s_%d" % (tag, H): npy(gv)})
    save("masks_layout", {"sizes": [32, 64], "ref": "sg2im/layout.py:48-77"}, **arrays)


def _graph_inputs(B, O, T, Din, Dp, P, seed):
    g = torch.Generator().manual_seed(seed)
    obj = torch.randn(B, O, Din, generator=g)
    pred = torch.randn(B, T, Dp, generator=g)
    edges = torch.randint(0, O, (B, T, 2), generator=g)
    p = torch.randint(1, P, (B, T), generator=g)
    tt = torch.randint(0, 4, (B, T), generator=g)
    p[0, -2:] = 0
    edges[0, -2:] = 0
    tt[0, -2:] = 0                                      # padded rows: [0, __padding__, 0], type 0
    p[1, -1] = 0
    edges[1, -1] = 0
    tt[1, -1] = 0
    edges[:, :, :][edges == O - 1] = 0                  # object O-1 has no incident triplet (count 0)
    return obj, pred, edges, p, tt


def fx_gconv():
    torch.manual_seed(1)
    P = 6
    w_trans = get_predicates_weights(P, "uniform")
    layer = GraphTripleConv(obj_input_dim=8, object_output_dim=20, predicate_input_dim=4,
                            predicate_output_dim=12, hidden_dim=16, num_attributes=1,
                            predicates_transitive_weights=w_trans)
    obj, pred, edges, p, tt = _graph_inputs(2, 5, 9, 8, 4, P, seed=11)
    obj.requires_grad_(True)
    pred.requires_grad_(True)
    new_obj, new_p = layer(obj, pred, edges, p != 0, tt, p)
    wo, wp = torch.randn_like(new_obj), torch.randn_like(new_p)
    ((new_obj * wo).sum() + (new_p * wp).sum()).backward()
    arrays = sd_np(layer, "sd:")
    arrays.update(grads_np(layer))
    arrays.update({"sd:predicates_transitive_weights": npy(w_trans), "grad:predicates_transitive_weights": npy(w_trans.grad),
                   "obj": npy(obj), "pred": npy(pred), "edges": npy(edges), "p": npy(p), "tt": npy(tt),
                   "new_obj": npy(new_obj), "new_p": npy(new_p), "wo": npy(wo), "wp": npy(wp),
                   "gobj": npy(obj.grad), "gpred": npy(pred.grad)})
    save("gconv", {"ref": "sg2im/graph.py:44-113", "hidden": 16, "dp_out": 12}, **arrays)


def fx_sg2layout():
    for kind, tag in (("tiny", "a1"), ("clevr", "a4")):
        torch.manual_seed(2)
        vocab = make_vocab(kind)
        opt = ref_opt(vocab, ["--embedding_dim", "8", "--gconv_dim", "16", "--gconv_hidden_dim", "24",
                              "--gconv_num_layers", "3", "--image_size", "32,32"])
        model = Sg2LayoutModel(opt)
        batch = make_batch(vocab, BatchConfig(3, 32, 2, 6, "packed"), seed=5)
        objs, boxes, triplets, tt = batch[1], batch[2], batch[3], batch[5].clone()
        tt[:, ::3] = 1                                   # some transitive edges
        tt[triplets[..., 1] == 0] = 0
        obj_vecs, boxes_pred, _ = model(objs, triplets, tt)
        wv, wb = torch.randn_like(obj_vecs), torch.randn_like(boxes_pred)
        ((obj_vecs * wv).sum() + (boxes_pred * wb).sum()).backward()
        arrays = sd_np(model, "sd:")
        arrays.update(grads_np(model))
        arrays.update({"objs": npy(objs), "triplets": npy(triplets), "tt": npy(tt), "obj_vecs": npy(obj_vecs),
                       "boxes_pred": npy(boxes_pred), "wv": npy(wv), "wb": npy(wb)})
        save("sg2layout_" + tag, {"ref": "sg2im/model.py:90-124", "vocab": kind,
                                  "argv": {"embedding_dim": 8, "gconv_dim": 16, "gconv_hidden_dim": 24,
                                           "gconv_num_layers": 3}}, **arrays)


def fx_spade_block():
    torch.manual_seed(3)
    vocab = make_vocab("tiny")
    opt = ref_opt(vocab, ["--embedding_dim", "4", "--image_size", "16,16"])
    blk = SPADEResnetBlock(16, 8, opt)                   # learned shortcut
    blk.load_state_dict(deterministic_state(blk.state_dict(), seed=3))
    blk.train()
    x = torch.randn(2, 16, 8, 8, requires_grad=True)
    seg = torch.randn(2, 4, 16, 16, requires_grad=True)
    y = blk(x, seg)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    arrays = sd_np(blk, "after:", skip=("weight_orig", "mlp_", "bias"))
    arrays.update(grads_np(blk))
    arrays.update({"x": npy(x), "seg": npy(seg), "y": npy(y), "w": npy(w), "gx": npy(x.grad), "gseg": npy(seg.grad)})
    # second call in eval mode: running stats, no power iteration
    blk.eval()
    arrays["y_eval"] = npy(blk(x, seg))
    save("spade_block", {"ref": "spade/models/networks/architecture.py:50-68", "fin": 16, "fout": 8,
                         "state": "deterministic_state(seed=3)", "embedding_dim": 4, "shapes": shapes_of(blk)}, **arrays)


def fx_syncbn():
    torch.manual_seed(4)
    bn = SynchronizedBatchNorm2d(5, affine=False)
    xs = [torch.randn(2, 5, 4, 4) * 2 + 1, torch.randn(2, 5, 4, 4) - 3]
    C = 5
    n = sum(x.shape[0] * 16 for x in xs)
    S = sum(x.transpose(0, 1).reshape(C, -1).sum(1) for x in xs)
    SS = sum((x.transpose(0, 1).reshape(C, -1) ** 2).sum(1) for x in xs)
    mean, inv_std = bn._compute_mean_std(S, SS, n)       # batchnorm.py:128-145 (the master's reduction)
    ys = [(x - mean.view(1, C, 1, 1)) * inv_std.view(1, C, 1, 1) for x in xs]
    # constant channel: variance 0 -> clamp(eps) path
    save("syncbn", {"ref": "spade/models/networks/sync_batchnorm/batchnorm.py:70-145"},
         x0=npy(xs[0]), x1=npy(xs[1]), y0=npy(ys[0]), y1=npy(ys[1]), mean=npy(mean), inv_std=npy(inv_std),
         running_mean=npy(bn.running_mean), running_var=npy(bn.running_var))


class _Holder(torch.nn.Module):
    def __init__(self, d, dobj=None, dmask=None):
        super().__init__()
        self.img_discriminator = d
        if dobj is not None:
            self.obj_discriminator = dobj
            self.mask_discriminator = dmask


def fx_crops():
    from sg2im.bilinear import crop_bbox_batch
    torch.manual_seed(8)
    vocab = make_vocab("tiny")
    batch = make_batch(vocab, BatchConfig(3, 32, 1, 4, "random"), seed=4)
    imgs, objs, boxes = batch[0].clone().requires_grad_(True), batch[1], batch[2].clone()
    boxes[0, 0] = torch.tensor([0.7, 0.6, 0.5, 0.6])           # runs off the image: zero padding
    crops = crop_bbox_batch(imgs, objs, boxes, 16, vocab=vocab)
    w = torch.randn_like(crops)
    (gi,) = torch.autograd.grad((crops * w).sum(), [imgs])
    save("crops", {"ref": "sg2im/bilinear.py:12-94", "vocab": "tiny", "size": 16}, imgs=npy(imgs), objs=npy(objs),
         boxes=npy(boxes), crops=npy(crops), w=npy(w), gimgs=npy(gi))


def fx_step_objdisc():
    """The DEFAULT trainer configuration (use_img_disc=0): image + object discriminators, both D steps."""
    torch.manual_seed(17)
    vocab = make_vocab("tiny")
    argv = ["--image_size", "64,64", "--embedding_dim", "8", "--gconv_dim", "16", "--gconv_hidden_dim", "24",
            "--gconv_num_layers", "2", "--ngf", "4", "--ndf", "4", "--no_vgg_loss", "--batch_size", "2",
            "--crop_size", "32", "--d_obj_arch", "C4-8-2,C4-16-2,C4-32-2"]
    opt = ref_opt(vocab, argv)
    assert not opt.use_img_disc
    sg, G, D = Sg2LayoutModel(opt), SPADEGenerator(opt), MultiscaleDiscriminator(opt)
    Dobj = AcCropDiscriminator(vocab=vocab, arch=opt.d_obj_arch, normalization=opt.d_normalization,
                               activation=opt.d_activation, padding=opt.d_padding, object_size=opt.crop_size)
    unused = ("repr_net", "image_encoder")
    sg.load_state_dict(deterministic_state(sg.state_dict(), seed=21))
    G.load_state_dict(deterministic_state(G.state_dict(), seed=22))
    D.load_state_dict(deterministic_state(D.state_dict(), seed=23))
    Dobj.load_state_dict(deterministic_state(Dobj.state_dict(), seed=24))
    batch = make_batch(vocab, BatchConfig(2, 64, 2, 5, "packed"), seed=19)
    imgs, objs, boxes, triplets, _, tt = batch[:6]
    arrays = {"imgs": npy(imgs), "objs": npy(objs), "boxes": npy(boxes), "triplets": npy(triplets), "tt": npy(tt)}
    gans = Pix2PixModel(opt, discriminator=_Holder(D, Dobj))
    for m in (sg, G, D, Dobj):
        m.train()
    trans = [p for n, p in sg.named_parameters() if n == "trans_candidates_weights"]
    base = [p for n, p in sg.named_parameters() if n not in ("trans_candidates_weights", "converse_candidates_weights")]
    base += list(G.parameters())
    optimizer = torch.optim.Adam([{"params": base, "lr": opt.learning_rate}, {"params": trans, "lr": 1e-2}])
    opt_d = torch.optim.Adam(list(D.parameters()), lr=opt.img_learning_rate, betas=(opt.beta1, 0.999))
    opt_o = torch.optim.Adam(list(Dobj.parameters()), lr=opt.learning_rate, betas=(opt.beta1, 0.999))
    _, boxes_pred, _ = sg(objs, triplets, tt, boxes)
    imgs_pred = G(objs, boxes, None, test_mode=False)
    model_out = (imgs_pred, boxes_pred, None)
    G_losses = gans(batch, model_out, mode="compute_generator_loss")
    for k, v in G_losses.items():
        arrays["G:" + k] = npy(v)
    optimizer.zero_grad()
    {k: v.mean() for k, v in G_losses.items()}["total_loss"].backward()
    for n, p in G.named_parameters():
        if p.grad is not None and n.endswith(("conv_img.weight", "up_3.conv_0.weight_orig", "head_0.norm_0.mlp_gamma.weight")):
            arrays["ggrad:" + n] = npy(p.grad)
    optimizer.step()
    D_losses = gans(batch, model_out, mode="compute_discriminator_loss")
    for k, v in D_losses.items():
        arrays["D:" + k] = npy(v)
    Dm = {k: v.mean() for k, v in D_losses.items()}
    opt_d.zero_grad(); Dm["total_img_loss"].backward(); opt_d.step()
    opt_o.zero_grad(); Dm["total_obj_loss"].backward()
    for n, p in Dobj.named_parameters():
        arrays["ograd:" + n] = npy(p.grad)
    opt_o.step()
    arrays.update(sd_np(Dobj, "o_after:"))
    for k, v in D.state_dict().items():
        if "weight_u" in k:
            arrays["d_after:" + k] = npy(v)              # 5 D_img calls advanced the power iteration 5x
    save("train_step_objdisc", {"ref": "scripts/train.py:353-393,468-485", "argv": argv, "vocab": "tiny",
                                "state": "deterministic_state seeds sg=21 g=22 d=23 dobj=24",
                                "shapes": {"sg": shapes_of(sg), "g": shapes_of(G, unused), "d": shapes_of(D, unused),
                                           "dobj": shapes_of(Dobj)}}, **arrays)


def fx_step_masks():
    """--mask_size 8: mask net, masks layout in G and D, mask BCE + mask discriminator terms, all four
    optimiser steps (scripts/train.py:353-393, 468-485)."""
    from spade.models.networks.discriminator import MultiscaleMaskDiscriminator2
    torch.manual_seed(27)
    vocab = make_vocab("tiny")
    argv = ["--image_size", "64,64", "--embedding_dim", "8", "--gconv_dim", "16", "--gconv_hidden_dim", "24",
            "--gconv_num_layers", "2", "--ngf", "4", "--ndf", "4", "--no_vgg_loss", "--batch_size", "2",
            "--crop_size", "32", "--d_obj_arch", "C4-8-2,C4-16-2,C4-32-2", "--mask_size", "8",
            "--g_mask_dim", "24", "--mask_noise_dim", "8", "--mask_pred_loss_weight", "0.5"]
    opt = ref_opt(vocab, argv)
    sg, G, D = Sg2LayoutModel(opt), SPADEGenerator(opt), MultiscaleDiscriminator(opt)
    Dobj = AcCropDiscriminator(vocab=vocab, arch=opt.d_obj_arch, normalization=opt.d_normalization,
                               activation=opt.d_activation, padding=opt.d_padding, object_size=opt.crop_size)
    Dmask = MultiscaleMaskDiscriminator2(opt)
    unused = ("repr_net", "image_encoder")
    sg.load_state_dict(deterministic_state(sg.state_dict(), seed=41))
    G.load_state_dict(deterministic_state(G.state_dict(), seed=42))
    D.load_state_dict(deterministic_state(D.state_dict(), seed=43))
    Dobj.load_state_dict(deterministic_state(Dobj.state_dict(), seed=44))
    Dmask.load_state_dict(deterministic_state(Dmask.state_dict(), seed=45))
    batch = make_batch(vocab, BatchConfig(2, 64, 2, 5, "packed", mask_size=8), seed=29)
    imgs, objs, boxes, triplets, _, tt, masks = batch[:7]
    arrays = {"imgs": npy(imgs), "objs": npy(objs), "boxes": npy(boxes), "triplets": npy(triplets), "tt": npy(tt),
              "masks": npy(masks)}
    gans = Pix2PixModel(opt, discriminator=_Holder(D, Dobj, Dmask))
    for m in (sg, G, D, Dobj, Dmask):
        m.train()
    trans = [p for n, p in sg.named_parameters() if n == "trans_candidates_weights"]
    base = [p for n, p in sg.named_parameters() if n not in ("trans_candidates_weights", "converse_candidates_weights")]
    base += list(G.parameters())
    optimizer = torch.optim.Adam([{"params": base, "lr": opt.learning_rate}, {"params": trans, "lr": 1e-2}])
    opt_d = torch.optim.Adam(list(D.parameters()), lr=opt.img_learning_rate, betas=(opt.beta1, 0.999))
    opt_o = torch.optim.Adam(list(Dobj.parameters()), lr=opt.learning_rate, betas=(opt.beta1, 0.999))
    opt_m = torch.optim.Adam(list(Dmask.parameters()), lr=opt.mask_learning_rate, betas=(opt.beta1, 0.999))
    torch.manual_seed(123)
    arrays["mask_noise"] = npy(torch.randn((1, opt.mask_noise_dim)))       # what create_mask_vecs draws next
    torch.manual_seed(123)
    _, boxes_pred, masks_pred = sg(objs, triplets, tt, boxes)
    arrays["masks_pred"] = npy(masks_pred)
    imgs_pred = G(objs, boxes, masks, test_mode=False)                     # meta_models.py:48: GT masks when given
    arrays["imgs_pred"] = npy(imgs_pred)
    model_out = (imgs_pred, boxes_pred, masks_pred)
    G_losses = gans(batch, model_out, mode="compute_generator_loss")
    for k, v in G_losses.items():
        arrays["G:" + k] = npy(v)
    optimizer.zero_grad()
    {k: v.mean() for k, v in G_losses.items()}["total_loss"].backward()
    for n, p in sg.named_parameters():
        if n.startswith("mask_net") and p.grad is not None:
            arrays["sggrad:" + n] = npy(p.grad)
    optimizer.step()
    D_losses = gans(batch, model_out, mode="compute_discriminator_loss")
    for k, v in D_losses.items():
        arrays["D:" + k] = npy(v)
    Dm = {k: v.mean() for k, v in D_losses.items()}
    opt_d.zero_grad(); Dm["total_img_loss"].backward(); opt_d.step()
    opt_o.zero_grad(); Dm["total_obj_loss"].backward(); opt_o.step()
    opt_m.zero_grad(); Dm["total_mask_loss"].backward()
    for n, p in Dmask.named_parameters():
        arrays["mgrad:" + n] = npy(p.grad)
    opt_m.step()
    for k, v in sg.state_dict().items():
        if k.startswith("mask_net") and ("running_" in k or "num_batches" in k):
            arrays["sg_after:" + k] = npy(v)
    save("train_step_masks", {"ref": "scripts/train.py:353-393,468-485; sg2im/model.py:67-88,118-123; "
                                     "spade/models/networks/discriminator.py:264-356",
                              "argv": argv, "vocab": "tiny",
                              "state": "deterministic_state seeds sg=41 g=42 d=43 dobj=44 dmask=45",
                              "shapes": {"sg": shapes_of(sg), "g": shapes_of(G, unused), "d": shapes_of(D, unused),
                                         "dobj": shapes_of(Dobj), "dmask": shapes_of(Dmask)}}, **arrays)


def fx_canon_graph():
    """Canonical graph construction of the packed datasets, by the reference's own functions:
    BaseDataset.add_location_triplets / add_dummy_triplets / add_learnt_triplets
    (sg2im/data/base_dataset.py:35-151) in the order of packed_clevr_dialog.py:205-209, then the padding
    of packed_clevr_collate_fn (packed_clevr_dialog.py:249-330)."""
    from sg2im.data.base_dataset import BaseDataset
    try:
        from sg2im.data.packed_clevr_dialog import packed_clevr_collate_fn
    except Exception as e:                                   # pragma: no cover
        raise SystemExit("cannot import the reference collate: %r" % (e,))
    vocab = make_vocab("clevr")
    rng = np.random.default_rng(2024)
    arrays, cases = {}, []
    for ci, (sizes, trans) in enumerate((((2, 3, 4, 6, 9, 14), 0), ((3, 5, 8, 12, 20, 33), 1), ((40, 17, 3), 1))):
        ds = BaseDataset()
        ds.vocab, ds.include_dummies = vocab, True
        ds.learned_transitivity, ds.learned_converse, ds.learned_symmetry = bool(trans), False, False
        batch = []
        for n in sizes:
            wh = rng.uniform(0.05, 0.6, size=(n, 2))
            xy = rng.uniform(0.0, 1.0, size=(n, 2)) * (1.0 - wh)
            if n >= 6:                                       # exact ties of centres / edges exercise the strict compares
                xy[1] = xy[0]; wh[1] = wh[0]
                xy[3, 0] = xy[2, 0]
            bx = [tuple(float(v) for v in r) for r in np.concatenate([xy, wh], axis=1)]
            centers = torch.FloatTensor([[x0 + 0.5 * w, y0 + 0.5 * h] for x0, y0, w, h in bx])   # packed_clevr_dialog.py:190-196
            boxes = torch.FloatTensor(bx + [[-1, -1, -1, -1]])
            objs = {a: torch.LongTensor(list(rng.integers(1, max(vocab["attributes"][a].values()) + 1, size=n)) + [0])
                    for a in vocab["attributes"]}
            triplets = []
            ds.add_location_triplets(boxes, centers, objs["shape"], triplets)
            ds.add_dummy_triplets(objs["shape"], triplets)
            triplets, conv_counts, ttype = ds.add_learnt_triplets(triplets, boxes.size(0))
            batch.append((torch.zeros(3, 4, 4), objs, boxes, torch.LongTensor(triplets), torch.LongTensor(conv_counts),
                          torch.LongTensor(ttype), None, len(batch), centers))
        out = packed_clevr_collate_fn(vocab, [b[:8] for b in batch])
        _, all_objs, all_boxes, all_triplets, _, all_tt, _, _ = out
        O = all_boxes.shape[1]
        cen = torch.zeros(len(sizes), O, 2)
        for b, n in enumerate(sizes):
            cen[b, :n] = batch[b][8]
        arrays.update({"c%d_objs" % ci: npy(all_objs), "c%d_boxes" % ci: npy(all_boxes), "c%d_triplets" % ci: npy(all_triplets),
                       "c%d_tt" % ci: npy(all_tt), "c%d_n" % ci: np.asarray([n + 1 for n in sizes], np.int64),
                       "c%d_counts" % ci: np.asarray([len(b[3]) for b in batch], np.int64), "c%d_centers" % ci: npy(cen)})
        cases.append({"sizes": list(sizes), "learned_transitivity": trans})
    save("canon_graph", {"ref": "sg2im/data/base_dataset.py:35-151; scripts/graphs_utils.py:15-100; "
                                "sg2im/data/packed_clevr_dialog.py:205-209,249-330",
                         "vocab": "clevr", "cases": cases,
                         "note": "centres are an input: float32 of x0 + 0.5*w evaluated in python floats, as "
                                 "packed_clevr_dialog.py:190-196 computes them before the boxes become a FloatTensor"},
         **arrays)


def fx_canon_converse():
    """Canonical graphs with `--learned_converse 1`: BaseDataset.add_learnt_triplets drawing converse edges through
    get_edge_converse_triplets (scripts/graphs_utils.py:126-152) from numpy's GLOBAL random stream, seeded here; the
    data loader's weights are numpy float32 (scripts/train.py:276).  Stored: the collated batch, conv_counts, the seed and
    the number of draws (the oracle and the HIP path replay the stream with np.random.seed(seed); random_sample(n))."""
    from sg2im.data.base_dataset import BaseDataset
    from sg2im.data.packed_clevr_dialog import packed_clevr_collate_fn
    vocab = make_vocab("clevr")
    rng = np.random.default_rng(77)
    arrays, cases = {}, []
    for ci, (sizes, trans, seed, scale) in enumerate((((2, 3, 5, 8, 13), 1, 11, 1.0), ((4, 9, 21, 34), 0, 12, 3.0),
                                                      ((30, 6, 18), 1, 13, 0.2))):
        ds = BaseDataset()
        ds.vocab, ds.include_dummies = vocab, True
        ds.learned_transitivity, ds.learned_converse, ds.learned_symmetry = bool(trans), True, False
        P = len(vocab["pred_name_to_idx"])
        w = torch.from_numpy(rng.normal(size=(P, P)).astype(np.float32) * scale)
        up = torch.triu(w, diagonal=0)
        ds.converse_candidates_weights = (up + up.t()).detach().cpu().numpy()          # model.py:10-13, train.py:276
        np.random.seed(seed)
        batch = []
        for n in sizes:
            wh = rng.uniform(0.05, 0.6, size=(n, 2))
            xy = rng.uniform(0.0, 1.0, size=(n, 2)) * (1.0 - wh)
            bx = [tuple(float(v) for v in r) for r in np.concatenate([xy, wh], axis=1)]
            centers = torch.FloatTensor([[x0 + 0.5 * ww, y0 + 0.5 * h] for x0, y0, ww, h in bx])
            boxes = torch.FloatTensor(bx + [[-1, -1, -1, -1]])
            objs = {a: torch.LongTensor(list(rng.integers(1, max(vocab["attributes"][a].values()) + 1, size=n)) + [0])
                    for a in vocab["attributes"]}
            triplets = []
            ds.add_location_triplets(boxes, centers, objs["shape"], triplets)
            ds.add_dummy_triplets(objs["shape"], triplets)
            triplets, conv_counts, ttype = ds.add_learnt_triplets(triplets, boxes.size(0))
            batch.append((torch.zeros(3, 4, 4), objs, boxes, torch.LongTensor(triplets), torch.FloatTensor(conv_counts),
                          torch.LongTensor(ttype), None, len(batch), centers))
        draws = int(sum(float(b[4].sum()) for b in batch))
        out = packed_clevr_collate_fn(vocab, [b[:8] for b in batch])
        _, all_objs, all_boxes, all_triplets, all_conv, all_tt, _, _ = out
        O = all_boxes.shape[1]
        cen = torch.zeros(len(sizes), O, 2)
        for b, n in enumerate(sizes):
            cen[b, :n] = batch[b][8]
        arrays.update({"c%d_objs" % ci: npy(all_objs), "c%d_boxes" % ci: npy(all_boxes), "c%d_triplets" % ci: npy(all_triplets),
                       "c%d_tt" % ci: npy(all_tt), "c%d_n" % ci: np.asarray([n + 1 for n in sizes], np.int64),
                       "c%d_counts" % ci: np.asarray([len(b[3]) for b in batch], np.int64), "c%d_centers" % ci: npy(cen),
                       "c%d_conv" % ci: npy(all_conv), "c%d_weights" % ci: ds.converse_candidates_weights.copy()})
        cases.append({"sizes": list(sizes), "learned_transitivity": trans, "seed": seed, "draws": draws})
    save("canon_converse", {"ref": "sg2im/data/base_dataset.py:89-139; scripts/graphs_utils.py:126-152; scripts/train.py:276",
                            "vocab": "clevr", "cases": cases}, **arrays)


def fx_converse():
    """REINFORCE signal of --learned_converse (scripts/train.py:343-345,370-378) from the reference's own
    calc_log_p / get_conv_converse on seeded inputs."""
    from scripts.graphs_utils import calc_log_p
    from sg2im.model import get_conv_converse
    vocab = make_vocab("coco")
    P = len(vocab["pred_idx_to_name"])
    g = torch.Generator().manual_seed(61)
    w = (torch.rand(P, P, generator=g) * 2 - 1).requires_grad_(True)
    counts = torch.randint(0, 4, (5, P, P + 1), generator=g).float()
    counts[:, :2] = 0                                                      # meta relations never sample
    r = torch.rand(5, generator=g) * 3
    meta = [vocab["pred_name_to_idx"][p] for p in ("__padding__", "__in_image__")]
    non_meta = set(vocab["pred_name_to_idx"].values()) - set(meta)
    eps = np.finfo(np.float32).eps.item()
    rn = (r - r.mean()) / (r.std() + eps)
    log_prob = calc_log_p(get_conv_converse({"sg_to_layout.module.converse_candidates_weights": w}), non_meta, counts)
    loss = torch.mean(rn * log_prob)
    loss.backward()
    save("converse", {"ref": "scripts/train.py:343-345,370-378; scripts/graphs_utils.py:109-123; sg2im/model.py:10-13",
                      "vocab": "coco"}, w=npy(w), counts=npy(counts), r=npy(r), log_prob=npy(log_prob), loss=npy(loss),
         grad_w=npy(w.grad))


def fx_model_and_step():
    """Sg2Layout + SPADEGenerator + MultiscaleDiscriminator at 64^2, ngf=2, ndf=4: forward
    outputs, loss dicts, gradients of the G step, and the state after one full train step
    (scripts/train.py:353-393 replayed by hand with the reference's own modules)."""
    torch.manual_seed(7)
    vocab = make_vocab("tiny")
    argv = ["--image_size", "64,64", "--embedding_dim", "8", "--gconv_dim", "16", "--gconv_hidden_dim", "24",
            "--gconv_num_layers", "2", "--ngf", "4", "--ndf", "4", "--no_vgg_loss", "--use_img_disc", "1",
            "--batch_size", "2"]
    opt = ref_opt(vocab, argv)
    sg = Sg2LayoutModel(opt)
    G = SPADEGenerator(opt)
    D = MultiscaleDiscriminator(opt)
    unused = ("repr_net", "image_encoder")               # never used in forward (generator.py:50-62)
    batch = make_batch(vocab, BatchConfig(2, 64, 2, 5, "packed"), seed=9)
    imgs, objs, boxes, triplets, _, tt = batch[:6]
    arrays = {"imgs": npy(imgs), "objs": npy(objs), "boxes": npy(boxes), "triplets": npy(triplets), "tt": npy(tt)}
    sg.load_state_dict(deterministic_state(sg.state_dict(), seed=11))
    G.load_state_dict(deterministic_state(G.state_dict(), seed=12))
    D.load_state_dict(deterministic_state(D.state_dict(), seed=13))

    gans = Pix2PixModel(opt, discriminator=_Holder(D))
    sg.train(); G.train(); D.train()
    trans = [p for n, p in sg.named_parameters() if n == "trans_candidates_weights"]
    base = [p for n, p in sg.named_parameters() if n not in ("trans_candidates_weights", "converse_candidates_weights")]
    base += list(G.parameters())
    optimizer = torch.optim.Adam([{"params": base, "lr": opt.learning_rate}, {"params": trans, "lr": 1e-2}])
    optimizer_d = torch.optim.Adam(list(D.parameters()), lr=opt.img_learning_rate, betas=(opt.beta1, 0.999))

    # --- forward (meta_models.py:25-51)
    obj_vecs, boxes_pred, _ = sg(objs, triplets, tt, boxes)
    imgs_pred = G(objs, boxes, None, test_mode=False)
    arrays.update({"boxes_pred": npy(boxes_pred), "imgs_pred": npy(imgs_pred)})
    model_out = (imgs_pred, boxes_pred, None)
    G_losses = gans(batch, model_out, mode="compute_generator_loss")
    for k, v in G_losses.items():
        arrays["G:" + k] = npy(v)
    G_mean = {k: v.mean() for k, v in G_losses.items()}
    optimizer.zero_grad()
    G_mean["total_loss"].backward()
    for n, p in G.named_parameters():
        if p.grad is not None and not any(u in n for u in unused):
            if n.endswith(("conv_img.weight", "fc.weight", "up_3.conv_0.weight_orig", "head_0.norm_0.mlp_gamma.weight",
                           "up_1.norm_s.mlp_shared.0.weight", "attribute_embedding.att_emb_0.weight")):
                arrays["ggrad:" + n] = npy(p.grad)
    for n, p in sg.named_parameters():
        if p.grad is not None:
            arrays["sggrad:" + n] = npy(p.grad)
    optimizer.step()
    # --- D step (train.py:390-393, :468-472)
    D_losses = gans(batch, model_out, mode="compute_discriminator_loss")
    for k, v in D_losses.items():
        arrays["D:" + k] = npy(v)
    D_mean = {k: v.mean() for k, v in D_losses.items()}
    optimizer_d.zero_grad()
    D_mean["total_img_loss"].backward()
    for n, p in D.named_parameters():
        if p.grad is not None and not any(u in n for u in unused):
            arrays["dgrad:" + n] = npy(p.grad)
    optimizer_d.step()
    arrays.update(sd_np(sg, "sg_after:"))
    arrays.update(sd_np(D, "d_after:", skip=unused))
    # G after the step: buffers fully, parameters as (sum, abs-sum) checksums + a few tensors
    for k, v in G.state_dict().items():
        if any(u in k for u in unused):
            continue
        if "running_" in k or "weight_u" in k or "weight_v" in k or "num_batches" in k or \
                k in ("conv_img.weight", "conv_img.bias", "fc.bias", "up_3.conv_1.weight_orig", "head_0.conv_0.bias"):
            arrays["g_after:" + k] = npy(v)
        else:
            arrays["g_after_sum:" + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    # D features of the real image (layout + concat + both scales) with the post-step weights, eval mode
    D.eval()
    with torch.no_grad():
        feats = D(imgs, objs, boxes)
    for i, scale in enumerate(feats):
        for j, f in enumerate(scale):
            arrays["dfeat_%d_%d" % (i, j)] = npy(f)
    save("train_step", {"ref": "scripts/train.py:353-393", "argv": argv, "vocab": "tiny",
                        "note": "use_img_disc=1, no VGG loss, learned_converse=0",
                        "state": "deterministic_state seeds sg=11 g=12 d=13",
                        "shapes": {"sg": shapes_of(sg), "g": shapes_of(G, unused), "d": shapes_of(D, unused)}}, **arrays)


def _tv_vgg19(pretrained=False):
    """Stand-in for torchvision.models.vgg19 (absent here): the public configuration 'E' feature stack
    with default-initialised weights — the fixture overwrites them by name afterwards."""
    import torch.nn as nn
    layers, cin = [], 3
    for v in (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M'):
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    m = nn.Module()
    m.features = nn.Sequential(*layers)
    return m


def fx_vgg():
    """VGGLoss / VGG19 of the reference (loss.py:102-117, architecture.py:93-123) on seeded weights."""
    import torch.nn as nn
    import spade.models.networks.architecture as ref_arch
    from spade.models.networks.loss import VGGLoss
    ref_arch.torchvision.models.vgg19 = _tv_vgg19
    cuda, nn.Module.cuda = nn.Module.cuda, (lambda self, *a, **k: self)      # VGGLoss calls .cuda() (loss.py:105)
    try:
        crit = VGGLoss([])
    finally:
        nn.Module.cuda = cuda
    crit.vgg.load_state_dict(deterministic_state(crit.vgg.state_dict(), seed=31))
    g = torch.Generator().manual_seed(77)
    x = (torch.rand(2, 3, 36, 44, generator=g) * 2 - 1).requires_grad_(True)
    y = torch.rand(2, 3, 36, 44, generator=g) * 2 - 1
    feats = crit.vgg(x)
    loss = crit(x, y)
    loss.backward()
    arrays = {"x": npy(x), "y": npy(y), "loss": npy(loss), "grad_x": npy(x.grad)}
    for i, f in enumerate(feats):
        arrays["feat_abs_mean_%d" % i] = npy(f.abs().mean())
        if i >= 2:
            arrays["feat_%d" % i] = npy(f)
    save("vgg_loss", {"ref": "spade/models/networks/loss.py:102-117, architecture.py:93-123",
                      "state": "deterministic_state seed 31 over the reference VGG19's state_dict",
                      "note": "torchvision is absent: vgg19() is the public configuration-E stack built in "
                              "make_golden.py; the slicing, loss weights and detach are the reference's",
                      "shapes": shapes_of(crit.vgg)}, **arrays)


def fx_row_gaps():
    """Reference outputs for the parts of rows a4, a7, a8, a12 that were missing after round 1: painter's
    compositing (`masks_to_layout(test_mode=True)`), box gradients of both layouts, non-square layouts,
    `build_mlp(batch_norm='batch')`, affine SynchronizedBatchNorm2d."""
    from sg2im.layers import build_mlp
    torch.manual_seed(61)
    arrays = {}
    # ---- painter's compositing, square and non-square, binary and soft masks (layout.py:71-74, 135-151)
    vecs = torch.randn(6, 8)
    boxes = torch.tensor([[0.10, 0.20, 0.50, 0.40], [0.05, 0.05, 0.90, 0.90], [0.60, 0.55, 0.35, 0.30],
                          [0.30, 0.30, 0.17, 0.60], [-0.2, 0.40, 0.60, 0.20], [0.45, 0.05, 0.50, 0.31]])
    yy, xx = torch.meshgrid((torch.arange(16) + 0.5) / 16, (torch.arange(16) + 0.5) / 16, indexing="ij")
    cen, rad = torch.rand(6, 2) * 0.3 + 0.35, torch.rand(6, 2) * 0.25 + 0.25
    ell = torch.stack([((xx - cen[i, 0]) / rad[i, 0]) ** 2 + ((yy - cen[i, 1]) / rad[i, 1]) ** 2 for i in range(6)])
    binm = (ell <= 1.0).long()
    soft = torch.sigmoid(4.0 * (1.0 - ell))                              # mask-net style probabilities
    arrays.update({"p_vecs": npy(vecs), "p_boxes": npy(boxes), "p_bin": npy(binm), "p_soft": npy(soft)})
    for (H, W) in ((32, 32), (24, 40)):
        for tag, m in (("bin", binm), ("soft", soft)):
            arrays["paint_%s_%dx%d" % (tag, H, W)] = npy(masks_to_layout(vecs, boxes, m, H, W, test_mode=True))
    # ---- box gradients of the masks layout (train mode), square and non-square
    v2 = vecs.clone().requires_grad_(True)
    b2 = boxes.clone().requires_grad_(True)
    for (H, W) in ((32, 32), (24, 40)):
        out = masks_to_layout(v2, b2, soft, H, W)
        w = torch.randn_like(out)
        gv, gb = torch.autograd.grad((out * w).sum(), [v2, b2])
        tag = "%dx%d" % (H, W)
        arrays.update({"m_out_" + tag: npy(out), "m_w_" + tag: npy(w), "m_gvecs_" + tag: npy(gv), "m_gboxes_" + tag: npy(gb)})
    # ---- build_mlp(batch_norm='batch') on a 2-D input (sg2im/layers.py:6-25): Linear, BatchNorm1d, ReLU, Linear, ReLU
    mlp = build_mlp([12, 24, 8], batch_norm='batch')
    mlp.load_state_dict(deterministic_state(mlp.state_dict(), seed=62))
    mlp.train()
    x = torch.randn(10, 12, requires_grad=True)
    y = mlp(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    arrays.update({"mlp_x": npy(x), "mlp_y": npy(y), "mlp_w": npy(w), "mlp_gx": npy(x.grad)})
    arrays.update(grads_np(mlp, "mlp_grad:"))
    arrays.update(sd_np(mlp, "mlp_after:"))
    mlp.eval()
    arrays["mlp_y_eval"] = npy(mlp(x))
    # ---- affine SynchronizedBatchNorm2d on one device (batchnorm.py:51-68): train step, then eval
    bn = SynchronizedBatchNorm2d(8)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(8) + 0.5)
        bn.bias.copy_(torch.randn(8) * 0.3)
    arrays.update({"bn_weight": npy(bn.weight), "bn_bias": npy(bn.bias)})
    bn.train()
    xb = (torch.randn(3, 8, 5, 6) * 1.5 + 0.4).requires_grad_(True)
    yb = bn(xb)
    wb = torch.randn_like(yb)
    (yb * wb).sum().backward()
    arrays.update({"bn_x": npy(xb), "bn_y": npy(yb), "bn_w": npy(wb), "bn_gx": npy(xb.grad), "bn_gweight": npy(bn.weight.grad),
                   "bn_gbias": npy(bn.bias.grad), "bn_running_mean": npy(bn.running_mean),
                   "bn_running_var": npy(bn.running_var)})
    bn.eval()
    arrays["bn_y_eval"] = npy(bn(xb))
    save("row_gaps", {"ref": "sg2im/layout.py:48-77,98-110,135-151; sg2im/layers.py:6-25; "
                             "spade/models/networks/sync_batchnorm/batchnorm.py:51-68",
                      "mlp_shapes": shapes_of(mlp)}, **arrays)


def fx_row_gaps_r3():
    """Round-3 row gaps: gradients w.r.t. the masks of masks_to_layout (layout.py:48-77) and w.r.t. the boxes of
    crop_bbox_batch (bilinear.py:44-94)."""
    from sg2im.bilinear import crop_bbox_batch
    torch.manual_seed(31)
    arrays = {}
    vecs = torch.randn(5, 8, requires_grad=True)
    boxes = torch.tensor([[0.10, 0.20, 0.50, 0.40], [0.00, 0.00, 1.00, 1.00], [0.60, 0.55, 0.35, 0.30],
                          [0.30, 0.30, 0.07, 0.90], [-0.2, 0.40, 0.60, 0.20]], requires_grad=True)
    for M in (16, 5):
        soft = torch.rand(5, M, M, requires_grad=True)
        arrays["soft_%d" % M] = npy(soft)
        for H, W in ((32, 32), (24, 40)):
            out = masks_to_layout(vecs, boxes, soft, H, W)
            w = torch.randn_like(out)
            gv, gb, gm = torch.autograd.grad((out * w).sum(), [vecs, boxes, soft])
            tag = "%d_%dx%d" % (M, H, W)
            arrays.update({"out_" + tag: npy(out), "w_" + tag: npy(w), "gvecs_" + tag: npy(gv), "gboxes_" + tag: npy(gb),
                           "gmasks_" + tag: npy(gm)})
    arrays.update({"vecs": npy(vecs), "boxes": npy(boxes)})
    # crops: gradients w.r.t. image AND boxes
    vocab = make_vocab("tiny")
    batch = make_batch(vocab, BatchConfig(3, 32, 1, 4, "random"), seed=9)
    imgs, objs = batch[0].clone().requires_grad_(True), batch[1]
    cb = batch[2].clone()
    cb[0, 0] = torch.tensor([0.7, 0.6, 0.5, 0.6])            # runs off the image: zero padding
    cb[1, 0] = torch.tensor([0.013, 0.027, 0.41, 0.33])
    cb = cb.requires_grad_(True)
    crops = crop_bbox_batch(imgs, objs, cb, 16, vocab=vocab)
    cw = torch.randn_like(crops)
    gi, gcb = torch.autograd.grad((crops * cw).sum(), [imgs, cb])
    arrays.update({"c_imgs": npy(imgs), "c_objs": npy(objs), "c_boxes": npy(cb), "c_crops": npy(crops), "c_w": npy(cw),
                   "c_gimgs": npy(gi), "c_gboxes": npy(gcb)})
    save("row_gaps_r3", {"ref": "sg2im/layout.py:48-77, sg2im/bilinear.py:12-94", "mask_sizes": [16, 5],
                         "sizes": [[32, 32], [24, 40]], "vocab": "tiny", "crop_size": 16}, **arrays)


def _buffers_np(module, prefix):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()
            if any(t in k for t in ("running_", "weight_u", "weight_v", "num_batches_tracked"))}


def fx_variants():
    """Off-recipe spellings of the builders, generated by the reference's own constructors (round 4; they were pinned to
    torch restatements written inside the tests before): `build_mlp` activations / dropout / batch norm
    (sg2im/layers.py:6-25), `build_cnn` grammar (sg2im/layers.py:28-112), `NLayerDiscriminator` under every `norm_D` the
    reference's `get_nonspade_norm_layer` can build (normalization.py:16-50 — its non-spectral spellings raise
    UnboundLocalError there, so only the four `spectral*` ones exist)."""
    import argparse
    from sg2im.layers import build_cnn, build_mlp
    from spade.models.networks.discriminator import NLayerDiscriminator
    arrays, meta = {}, {"ref": "sg2im/layers.py:6-112; spade/models/networks/normalization.py:16-50, discriminator.py:163-206",
                        "mlp": [], "cnn": [], "nld": []}
    # ---- build_mlp
    mlp_cases = [dict(activation='leakyrelu-0.2', final_nonlinearity=None),
                 dict(activation='sigmoid', final_nonlinearity='sigmoid'),
                 dict(batch_norm='batch', activation='leakyrelu-0.3', final_nonlinearity='leakyrelu'),
                 dict(dropout=0.5, final_nonlinearity='relu')]
    dims = [12, 32, 24, 8]
    for i, kw in enumerate(mlp_cases):
        torch.manual_seed(70 + i)
        m = build_mlp(dims, **kw)
        m.load_state_dict(deterministic_state(m.state_dict(), seed=70 + i))
        train = kw.get('dropout', 0) == 0                    # dropout masks of two RNGs cannot agree: eval mode there
        m.train(train)
        x = torch.randn(40, 12, requires_grad=True)
        w = torch.randn(40, 8)
        y = m(x)
        (y * w).sum().backward()
        tag = "mlp%d_" % i
        arrays.update({tag + "x": npy(x), tag + "w": npy(w), tag + "y": npy(y), tag + "gx": npy(x.grad)})
        arrays.update(grads_np(m, tag + "grad:"))
        arrays.update(_buffers_np(m, tag + "after:"))
        meta["mlp"].append({"kw": kw, "dims": dims, "seed": 70 + i, "train": train, "shapes": shapes_of(m),
                            "keys": list(m.state_dict().keys()), "len": len(m)})
    # ---- build_cnn
    cnn_cases = [('I8,C3-16,C3-32-2,U2,C3-8,P2,C1-4', dict(normalization='instance', activation='relu'), (2, 8, 16, 16)),
                 ('I8,C3-16,C3-32', dict(normalization='none', activation='leakyrelu-0.1'), (2, 8, 12, 12)),
                 ('C4-16-2,C4-32-2', dict(normalization='batch', activation='sigmoid', padding='valid'), (3, 3, 22, 22)),
                 ('I4,C3-8,U3,C3-8', dict(normalization='batch', activation='leakyrelu-0.2'), (2, 4, 8, 8)),
                 # round 4: residual blocks (the reference evaluates their net twice per call), average pooling, FC layers
                 ('I8,C3-16,R,P2,C3-8', dict(normalization='batch', activation='relu', pooling='avg'), (3, 8, 12, 12)),
                 ('I4,R,C3-8-2,R,FC-128-12,FC-12-4', dict(normalization='instance', activation='leakyrelu-0.2'), (2, 4, 8, 8)),
                 ('I8,R,P2,FC-128-8', dict(normalization='none', activation='relu', pooling='avg'), (3, 8, 8, 8))]
    for i, (arch, kw, shape) in enumerate(cnn_cases):
        torch.manual_seed(80 + i)
        m, cout = build_cnn(arch, **kw)
        m.load_state_dict(deterministic_state(m.state_dict(), seed=80 + i))
        m.train()
        x = torch.randn(*shape, requires_grad=True)
        y = m(x)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        tag = "cnn%d_" % i
        arrays.update({tag + "x": npy(x), tag + "w": npy(w), tag + "y": npy(y), tag + "gx": npy(x.grad)})
        arrays.update(grads_np(m, tag + "grad:"))
        arrays.update(_buffers_np(m, tag + "after:"))
        meta["cnn"].append({"arch": arch, "kw": kw, "seed": 80 + i, "cout": cout, "shapes": shapes_of(m),
                            "keys": list(m.state_dict().keys()), "len": len(m)})
    # ---- NLayerDiscriminator under the norm_D spellings the reference can build
    for i, norm_D in enumerate(["spectralinstance", "spectralbatch", "spectralsync_batch", "spectralnone"]):
        opt = argparse.Namespace(ndf=8, n_layers_D=4, norm_D=norm_D, semantic_nc=5, no_ganFeat_loss=False)
        torch.manual_seed(90 + i)
        D = NLayerDiscriminator(opt)
        D.load_state_dict(deterministic_state(D.state_dict(), seed=90 + i))
        D.train()
        x = torch.randn(2, 8, 36, 36, requires_grad=True)
        feats = D(x)
        w = torch.randn_like(feats[-1])
        (feats[-1] * w).sum().backward()
        tag = "nld%d_" % i
        arrays.update({tag + "x": npy(x), tag + "w": npy(w), tag + "gx": npy(x.grad)})
        for j, f in enumerate(feats):
            arrays[tag + "feat%d" % j] = npy(f)
        arrays.update(grads_np(D, tag + "grad:"))
        arrays.update(_buffers_np(D, tag + "after:"))            # u / v after the power iteration, running statistics
        meta["nld"].append({"norm_D": norm_D, "seed": 90 + i, "shapes": shapes_of(D), "keys": list(D.state_dict().keys()),
                            "n_feats": len(feats)})
    save("variants", meta, **arrays)


if __name__ == "__main__":
    torch.set_num_threads(4)
    if len(sys.argv) > 1:                      # regenerate selected fixtures: make_golden.py fx_vgg ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    fx_layout()
    fx_masks_layout()
    fx_gconv()
    fx_sg2layout()
    fx_spade_block()
    fx_syncbn()
    fx_model_and_step()
    fx_crops()
    fx_step_objdisc()
    fx_vgg()
    fx_step_masks()
    fx_canon_graph()
    fx_canon_converse()
    fx_converse()
    fx_row_gaps()
    fx_row_gaps_r3()
    fx_variants()
