"""Functional CPU restatement of the reference hot path (see package docstring).

Conventions
-----------
* `state` is a flat dict {reference state_dict key: tensor}.  Parameters may
  require grad; buffers (`running_mean`, `weight_u`, ...) are updated IN PLACE
  exactly where the reference's modules mutate them.
* `prefix` selects a sub-module ("gconvs.0.", "up_3.norm_s.", ...).
* All citations are relative to /root/reference.
"""

import torch
import torch.nn.functional as F

__all__ = [
    "attribute_embeddings", "mlp2", "triplet_confidence", "graph_triple_conv", "sg2layout_forward",
    "remove_dummy_objects", "box_coverage", "boxes_to_layout", "masks_to_layout", "batched_layout",
    "spectral_weight", "batch_norm_train", "syncbn_multi_replica", "spade", "spade_resblock",
    "generator_forward", "instance_norm", "nlayer_discriminator", "multiscale_discriminator",
    "hinge_loss", "gan_loss_multiscale", "generator_losses", "discriminator_losses", "TrainState",
    "train_step", "make_adam_groups", "crop_objects", "ac_crop_discriminator", "bce_loss", "vgg19_features", "vgg_loss", "mask_net", "mask_discriminator", "calc_log_p", "converse_loss",
    "mlp2_batchnorm", "syncbn_affine_single_device",
]

ORIGINAL_EDGE, TRANSITIVE_EDGE = 0, 1          # sg2im/data/base_dataset.py:7-8


# --------------------------------------------------------------------------- graph encoder
def attribute_embeddings(state, prefix, objs):
    """`AttributeEmbeddings.forward` (sg2im/attribute_embed.py:31-48): one table per attribute
    column, concatenated; the Linear exists iff A > 1 or use_attr_fc_gen (ctor :24-25)."""
    cols = [F.embedding(objs[..., k], state["%satt_emb_%d.weight" % (prefix, k)])
            for k in range(objs.shape[-1])]
    v = torch.cat(cols, dim=-1)
    wkey = prefix + "attribute_fc_gen.weight"
    if wkey in state:
        v = F.linear(v, state[wkey], state[prefix + "attribute_fc_gen.bias"])
    return v


def mlp2(state, prefix, x, final_relu):
    """`build_mlp([d0,d1,d2])` with mlp_normalization='none' (sg2im/layers.py:6-25):
    Linear(net.0) ReLU Linear(net.2) [ReLU]."""
    h = F.relu(F.linear(x, state[prefix + "0.weight"], state[prefix + "0.bias"]))
    y = F.linear(h, state[prefix + "2.weight"], state[prefix + "2.bias"])
    return F.relu(y) if final_relu else y


def mlp2_batchnorm(state, prefix, x, training, final_relu=True):
    """`build_mlp([d0,d1,d2], batch_norm='batch')` (sg2im/layers.py:6-25): Linear(net.0) BatchNorm1d(net.1) ReLU
    Linear(net.3) [ReLU].  nn.BatchNorm1d normalises dim 1, so the input is (N, d0) (or (N, d0, L))."""
    h = F.linear(x, state[prefix + "0.weight"], state[prefix + "0.bias"])
    if training:
        state[prefix + "1.num_batches_tracked"].add_(1)
    h = F.batch_norm(h, state[prefix + "1.running_mean"], state[prefix + "1.running_var"], state[prefix + "1.weight"],
                     state[prefix + "1.bias"], training, 0.1, 1e-5)
    y = F.linear(F.relu(h), state[prefix + "3.weight"], state[prefix + "3.bias"])
    return F.relu(y) if final_relu else y


def syncbn_affine_single_device(state, prefix, x, training, momentum=0.1, eps=1e-5):
    """`_SynchronizedBatchNorm.forward` with affine=True on one device (sync_batchnorm/batchnorm.py:63-68):
    F.batch_norm with weight and bias; `num_batches_tracked` is not advanced (the forward bypasses _BatchNorm's)."""
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"], state[prefix + "weight"],
                        state[prefix + "bias"], training, momentum, eps)


def triplet_confidence(w_trans, triplet_type, predicate_ids):
    """sg2im/graph.py:69-74: 1 for original edges, sigmoid(w[p]) for transitive ones, 0 otherwise."""
    tt = triplet_type.to(w_trans.dtype)
    sig = torch.sigmoid(w_trans)
    return (tt == ORIGINAL_EDGE).to(w_trans.dtype) + (tt == TRANSITIVE_EDGE).to(w_trans.dtype) * sig[predicate_ids]


def graph_triple_conv(state, prefix, obj_vecs, pred_vecs, edges, pred_indicators, triplet_type,
                      predicate_ids, w_trans):
    """`GraphTripleConv.forward` (sg2im/graph.py:44-113) in batched scatter form.

    The per-sample python loop of :85-107 is replaced by one masked scatter_add per role;
    SURVEY.md appendix A records that this form is bit-identical on CPU.  Pooling is always an
    average whose divisor is the SUM OF CONFIDENCES (:101-106); padded triplets
    (pred_indicators False) still pass through net1 and produce new_p rows (:67, :80)."""
    B, O, _ = obj_vecs.shape
    H = state[prefix + "net2.0.weight"].shape[1]                 # hidden_dim
    Dp = state[prefix + "net1.2.weight"].shape[0] - 2 * H        # predicate_output_dim
    s_idx, o_idx = edges[..., 0], edges[..., 1]
    gather = lambda idx: torch.gather(obj_vecs, 1, idx.unsqueeze(-1).expand(-1, -1, obj_vecs.shape[-1]))
    cur_t = torch.cat([gather(s_idx), pred_vecs, gather(o_idx)], dim=-1)          # :63-66
    new_t = mlp2(state, prefix + "net1.", cur_t, final_relu=True)                 # :67
    conf = triplet_confidence(w_trans, triplet_type, predicate_ids)               # :69-74
    new_t = new_t * conf.unsqueeze(-1)                                            # :76-77
    new_s, new_p, new_o = new_t[..., :H], new_t[..., H:H + Dp], new_t[..., H + Dp:]
    valid = pred_indicators.to(new_t.dtype)
    exp = lambda idx: idx.unsqueeze(-1).expand(-1, -1, H)
    pooled = torch.zeros(B, O, H, dtype=new_t.dtype)
    pooled = pooled.scatter_add(1, exp(s_idx), new_s * valid.unsqueeze(-1))       # :98
    pooled = pooled.scatter_add(1, exp(o_idx), new_o * valid.unsqueeze(-1))       # :99
    cnt = torch.zeros(B, O, dtype=new_t.dtype)
    cnt = cnt.scatter_add(1, s_idx, conf * valid).scatter_add(1, o_idx, conf * valid)   # :101-103
    pooled = torch.where((cnt > 0).unsqueeze(-1), pooled / cnt.clamp_min(1e-38).unsqueeze(-1), pooled)  # :105-106
    new_obj = mlp2(state, prefix + "net2.", pooled, final_relu=True)              # :110
    return new_obj, new_p


def mask_net(state, prefix, mask_vecs, training=True):
    """`Sg2LayoutModel.mask_net` (sg2im/model.py:67-79): [nearest x2, Conv3x3, BatchNorm2d, ReLU]* then
    Conv1x1 -> 1, over EVERY (sample, slot) pair incl. padded slots (model.py:121)."""
    x = mask_vecs.reshape(-1, mask_vecs.shape[-1], 1, 1)
    i = 0
    while (prefix + "%d.running_mean" % (i + 2)) in state:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.conv2d(x, state[prefix + "%d.weight" % (i + 1)], state[prefix + "%d.bias" % (i + 1)], padding=1)
        x = F.relu(_batch_norm_affine(state, prefix + "%d." % (i + 2), x, training))
        i += 4
    return F.conv2d(x, state[prefix + "%d.weight" % i], state[prefix + "%d.bias" % i])


def sg2layout_forward(state, vocab, objs, triplets, triplet_type, prefix="", mask_noise=None):
    """`Sg2LayoutModel.forward` (sg2im/model.py:90-124).  The mask branch (:118-123) runs when the
    state holds a mask net; `mask_noise` is the (1, mask_noise_dim) row `create_mask_vecs` draws
    with torch.randn (:85) — passed in so that both sides of a comparison use the same draw."""
    s, p, o = triplets[..., 0], triplets[..., 1], triplets[..., 2]
    edges = torch.stack([s, o], dim=-1)
    pred_indicators = p != vocab["pred_name_to_idx"]["__padding__"]                # :107
    obj_vecs = attribute_embeddings(state, prefix + "attribute_embedding.", objs)  # :108
    pred_vecs = F.embedding(p, state[prefix + "pred_embeddings.weight"])           # :109
    w_trans = state[prefix + "trans_candidates_weights"]
    i = 0
    while (prefix + "gconvs.%d.net1.0.weight" % i) in state:                       # :111-112
        obj_vecs, pred_vecs = graph_triple_conv(state, prefix + "gconvs.%d." % i, obj_vecs, pred_vecs,
                                                edges, pred_indicators, triplet_type, p, w_trans)
        i += 1
    boxes_pred = mlp2(state, prefix + "box_net.", obj_vecs, final_relu=False)      # :115
    masks_pred = None
    if (prefix + "mask_net.1.weight") in state:                                    # :118-123
        B, O = objs.shape[0], objs.shape[1]
        noise = mask_noise.repeat((B, O, 1)).view(B, O, -1)                        # :85-86
        scores = mask_net(state, prefix + "mask_net.", torch.cat([obj_vecs, noise], dim=-1))
        masks_pred = scores.view(B, O, scores.shape[2], scores.shape[3]).sigmoid()
    return obj_vecs, boxes_pred, masks_pred


# --------------------------------------------------------------------------- layout
def remove_dummy_objects(objs_one, vocab):
    """sg2im/utils.py:56-63 — real objects of ONE sample; only attribute column 0 is tested."""
    col = objs_one[:, 0]
    return (col != 0) & (col != vocab["object_name_to_idx"]["__image__"])


def box_coverage(lo, size, n):
    """Separable weight of one box along one axis, (O,n).

    Closed form of `_boxes_to_grid` (sg2im/layout.py:80-112) followed by
    `F.grid_sample(bilinear, zeros, align_corners=False)` of a CONSTANT 8-pixel line
    (layout.py:34-35): pixel centres t=linspace(0,1,n); g=2*(t-lo)/size-1;
    ix=((g+1)*8-1)/2; weight = (1-frac)*[0<=i0<=7] + frac*[0<=i0+1<=7]."""
    t = torch.linspace(0, 1, steps=n, dtype=lo.dtype).view(1, n)
    g = ((t - lo.view(-1, 1)) / size.view(-1, 1)).mul(2).sub(1)
    ix = ((g + 1) * 8 - 1) / 2
    i0 = torch.floor(ix)
    fr = ix - i0
    in0 = ((i0 >= 0) & (i0 <= 7)).to(lo.dtype)
    in1 = ((i0 + 1 >= 0) & (i0 + 1 <= 7)).to(lo.dtype)
    return (1 - fr) * in0 + fr * in1


def boxes_to_layout(vecs, boxes, H, W=None):
    """`boxes_to_layout` (sg2im/layout.py:12-45) -> (1,D,H,W); boxes are [x0,y0,w,h] (:95-96).
    grid_sample of a constant image factorises: sampled[o,d,h,w] = vec[o,d]*cy[o,h]*cx[o,w];
    `_pool_samples` (:156-188) sums objects (obj_to_img is all zeros)."""
    W = H if W is None else W
    cx = box_coverage(boxes[:, 0], boxes[:, 2], W)
    cy = box_coverage(boxes[:, 1], boxes[:, 3], H)
    out = torch.zeros(vecs.shape[1], H, W, dtype=vecs.dtype)
    for o in range(vecs.shape[0]):                  # index order, like scatter_add on CPU
        out = out + vecs[o].view(-1, 1, 1) * (cy[o].view(1, H, 1) * cx[o].view(1, 1, W))
    return out.unsqueeze(0)


def _axis_taps(lo, size, n_out, n_src):
    """(i0, w0, w1): lower tap index and the two tap weights (zeroed outside the source) of the
    bilinear grid_sample(align_corners=False, zeros) along one axis — box_coverage generalised."""
    t = torch.linspace(0, 1, steps=n_out, dtype=lo.dtype).view(1, n_out)
    g = ((t - lo.view(-1, 1)) / size.view(-1, 1)).mul(2).sub(1)
    ix = ((g + 1) * n_src - 1) / 2
    i0 = torch.floor(ix)
    fr = ix - i0
    w0 = (1 - fr) * ((i0 >= 0) & (i0 <= n_src - 1)).to(lo.dtype)
    w1 = fr * ((i0 + 1 >= 0) & (i0 + 1 <= n_src - 1)).to(lo.dtype)
    return i0.clamp(-1, n_src - 1).long(), w0, w1


def _mask_samples(boxes, masks, H, W, dtype):
    """(O,H,W): bilinear sample of each object's (M,M) mask over its box — `F.grid_sample(masks, grid)` of
    sg2im/layout.py:70-73 (align_corners=False, zeros padding) in closed form."""
    O, M = masks.shape[0], masks.shape[1]
    mk = F.pad(masks.to(dtype), (1, 1, 1, 1))                       # index -1 and M read zeros
    ix0, wx0, wx1 = _axis_taps(boxes[:, 0], boxes[:, 2], W, M)
    iy0, wy0, wy1 = _axis_taps(boxes[:, 1], boxes[:, 3], H, M)
    out = []
    for o in range(O):
        r0, r1 = mk[o][iy0[o] + 1], mk[o][iy0[o] + 2]               # (H, M+2) source rows of each output row
        c0, c1 = ix0[o] + 1, ix0[o] + 2
        out.append(wy0[o].view(H, 1) * (r0[:, c0] * wx0[o].view(1, W) + r0[:, c1] * wx1[o].view(1, W)) +
                   wy1[o].view(H, 1) * (r1[:, c0] * wx0[o].view(1, W) + r1[:, c1] * wx1[o].view(1, W)))
    return torch.stack(out) if out else torch.zeros(0, H, W, dtype=dtype)


def masks_to_layout(vecs, boxes, masks, H, W=None, test_mode=False):
    """`masks_to_layout` (sg2im/layout.py:48-77): vec[o] x bilinear sample of mask[o] over box o.  Train mode sums
    the objects; `test_mode` composites them with the painter's algorithm of `_pool_mask_samples` (:135-151):
    ascending order of mass = sum(samples[j]); a pixel goes to the first object whose sampled mask is > 0.5.
    masks (O,M,M) int or float."""
    W = H if W is None else W
    w = _mask_samples(boxes, masks, H, W, vecs.dtype)               # (O,H,W)
    out = torch.zeros(vecs.shape[1], H, W, dtype=vecs.dtype)
    if not test_mode:
        for o in range(vecs.shape[0]):
            out = out + vecs[o].view(-1, 1, 1) * w[o].unsqueeze(0)
        return out.unsqueeze(0)
    samples = vecs.view(vecs.shape[0], -1, 1, 1) * w.unsqueeze(1)   # (O,D,H,W)
    mass = [float(samples[j].sum()) for j in range(vecs.shape[0])]
    taken = torch.zeros(H, W, dtype=vecs.dtype)
    for j in sorted(range(len(mass)), key=lambda i: mass[i]):       # np.argsort (:141); stable for ties
        claim = (taken == 0).to(vecs.dtype) * (w[j] > 0.5).to(vecs.dtype)
        taken = taken + claim
        out = out + samples[j] * claim
    return out.unsqueeze(0)


def batched_layout(obj_vecs, objs, boxes, vocab, H, masks=None):
    """The per-sample loop of spade/models/networks/generator.py:82-96 and discriminator.py:102-119
    (masks layout when `masks` is given, else boxes layout)."""
    segs = []
    for b in range(obj_vecs.shape[0]):
        m = remove_dummy_objects(objs[b], vocab)
        if masks is not None:
            segs.append(masks_to_layout(obj_vecs[b][m], boxes[b][m], masks[b][m], H, H))
        else:
            segs.append(boxes_to_layout(obj_vecs[b][m], boxes[b][m], H, H))
    return torch.cat(segs, dim=0)


# --------------------------------------------------------------------------- generator
def spectral_weight(state, prefix, training, eps=1e-12):
    """torch.nn.utils.spectral_norm pre-forward hook (call sites architecture.py:36-39,
    normalization.py:27): one power iteration per TRAINING-mode call on W.view(out,-1);
    u,v buffers persist (updated in place); W_eff = W_orig / (u^T W v)."""
    w = state[prefix + "weight_orig"]
    u, v = state[prefix + "weight_u"], state[prefix + "weight_v"]
    wm = w.reshape(w.shape[0], -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
    sigma = torch.dot(u.clone(), torch.mv(wm, v.clone()))
    return w / sigma


def batch_norm_train(x, running_mean, running_var, nbt, training, momentum=0.1, eps=1e-5):
    """1-device path of `_SynchronizedBatchNorm.forward` (sync_batchnorm/batchnorm.py:65-68) =
    F.batch_norm(affine=False): biased variance + eps inside the sqrt; running_var unbiased.
    `num_batches_tracked` is NOT advanced: that forward bypasses `_BatchNorm.forward`."""
    return F.batch_norm(x, running_mean, running_var, None, None, training, momentum, eps)


def syncbn_multi_replica(x_shards, running_mean, running_var, momentum=0.1, eps=1e-5):
    """N-device path (batchnorm.py:70-93, 128-145): sum and square-sum over all replicas,
    mean = S/n, var_b = (SS - S*mean)/n, inv_std = clamp(var_b, eps)^-1/2 (CLAMP, not +eps),
    running_var from the unbiased variance.  Returns the per-shard outputs."""
    C = x_shards[0].shape[1]
    n = sum(x.shape[0] * x[0, 0].numel() for x in x_shards)
    S = sum(x.transpose(0, 1).reshape(C, -1).sum(1) for x in x_shards)
    SS = sum((x.transpose(0, 1).reshape(C, -1) ** 2).sum(1) for x in x_shards)
    mean = S / n
    sumvar = SS - S * mean
    with torch.no_grad():
        running_mean.mul_(1 - momentum).add_(momentum * mean.detach())
        running_var.mul_(1 - momentum).add_(momentum * (sumvar / (n - 1)).detach())
    inv_std = (sumvar / n).clamp(eps) ** -0.5
    return [(x - mean.view(1, C, 1, 1)) * inv_std.view(1, C, 1, 1) for x in x_shards]


def spade(state, prefix, x, seg, training):
    """`SPADE.forward` (spade/models/networks/normalization.py:96-110)."""
    normalized = batch_norm_train(x, state[prefix + "param_free_norm.running_mean"],
                                  state[prefix + "param_free_norm.running_var"],
                                  state.get(prefix + "param_free_norm.num_batches_tracked"), training)
    segr = F.interpolate(seg, size=x.shape[2:], mode="nearest")                               # :102
    actv = F.relu(F.conv2d(segr, state[prefix + "mlp_shared.0.weight"], state[prefix + "mlp_shared.0.bias"],
                           padding=1))                                                        # :103
    gamma = F.conv2d(actv, state[prefix + "mlp_gamma.weight"], state[prefix + "mlp_gamma.bias"], padding=1)
    beta = F.conv2d(actv, state[prefix + "mlp_beta.weight"], state[prefix + "mlp_beta.bias"], padding=1)
    return normalized * (1 + gamma) + beta                                                    # :108


def spade_resblock(state, prefix, x, seg, training):
    """`SPADEResnetBlock.forward` (spade/models/networks/architecture.py:50-68)."""
    if (prefix + "conv_s.weight_orig") in state:                                              # learned shortcut
        w_s = spectral_weight(state, prefix + "conv_s.", training)
        x_s = F.conv2d(spade(state, prefix + "norm_s.", x, seg, training), w_s)               # :60-61
    else:
        x_s = x
    w0 = spectral_weight(state, prefix + "conv_0.", training)
    dx = F.conv2d(F.leaky_relu(spade(state, prefix + "norm_0.", x, seg, training), 0.2), w0,
                  state[prefix + "conv_0.bias"], padding=1)                                   # :53
    w1 = spectral_weight(state, prefix + "conv_1.", training)
    dx = F.conv2d(F.leaky_relu(spade(state, prefix + "norm_1.", dx, seg, training), 0.2), w1,
                  state[prefix + "conv_1.bias"], padding=1)                                   # :54
    return x_s + dx


def generator_forward(state, vocab, image_size, objs, layout_boxes, training=True, prefix="",
                      num_upsampling_layers="normal", layout_masks=None):
    """`SPADEGenerator.forward` (spade/models/networks/generator.py:79-127)."""
    H = image_size
    n_up = {"normal": 5, "more": 6, "most": 7}[num_upsampling_layers]                         # :64-77
    sw = H // (2 ** n_up)
    obj_vecs = attribute_embeddings(state, prefix + "attribute_embedding.", objs)             # :80
    seg = batched_layout(obj_vecs, objs, layout_boxes, vocab, H, layout_masks)                # :82-96
    x = F.interpolate(seg, size=(sw, sw))                                                     # :99
    x = F.conv2d(x, state[prefix + "fc.weight"], state[prefix + "fc.bias"], padding=1)        # :100
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    x = spade_resblock(state, prefix + "head_0.", x, seg, training)
    x = up(x)
    x = spade_resblock(state, prefix + "G_middle_0.", x, seg, training)
    if num_upsampling_layers in ("more", "most"):
        x = up(x)
    x = spade_resblock(state, prefix + "G_middle_1.", x, seg, training)
    for name in ("up_0.", "up_1.", "up_2.", "up_3."):
        x = up(x)
        x = spade_resblock(state, prefix + name, x, seg, training)
    if num_upsampling_layers == "most":
        x = up(x)
        x = spade_resblock(state, prefix + "up_4.", x, seg, training)
    x = F.conv2d(F.leaky_relu(x, 0.2), state[prefix + "conv_img.weight"], state[prefix + "conv_img.bias"],
                 padding=1)                                                                   # :123
    return torch.tanh(x)                                                                      # :124


# --------------------------------------------------------------------------- discriminator
def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False) (normalization.py:44)."""
    return F.instance_norm(x, eps=eps)


def nlayer_discriminator(state, prefix, x, training):
    """`NLayerDiscriminator.forward` (spade/models/networks/discriminator.py:164-206): conv4x4 s2
    p2 + LReLU; (n_layers-1) x [SN-conv4x4 (no bias) + IN + LReLU], last of them stride 1;
    conv4x4 -> 1.  Returns every block output (feature matching)."""
    outs = []
    h = F.leaky_relu(F.conv2d(x, state[prefix + "model0.0.weight"], state[prefix + "model0.0.bias"],
                              stride=2, padding=2), 0.2)
    outs.append(h)
    n = 1
    while (prefix + "model%d.0.0.weight_orig" % n) in state:
        last = (prefix + "model%d.0.0.weight_orig" % (n + 1)) not in state
        w = spectral_weight(state, prefix + "model%d.0.0." % n, training)
        h = F.leaky_relu(instance_norm(F.conv2d(h, w, None, stride=1 if last else 2, padding=2)), 0.2)
        outs.append(h)
        n += 1
    h = F.conv2d(h, state[prefix + "model%d.0.weight" % n], state[prefix + "model%d.0.bias" % n],
                 stride=1, padding=2)
    outs.append(h)
    return outs


def multiscale_discriminator(state, vocab, image_size, img, objs, layout_boxes, training=True, prefix="",
                             layout_masks=None):
    """`MultiscaleDiscriminator.forward` (discriminator.py:97-131): D's own embedding (with fc,
    :71-72) -> layout -> cat(img, layout) -> num_D scales, avg_pool(3,s2,p1, no pad count) between."""
    obj_vecs = attribute_embeddings(state, prefix + "attribute_embedding.", objs)
    seg = batched_layout(obj_vecs, objs, layout_boxes, vocab, image_size, layout_masks)
    inp = torch.cat([img, seg], dim=1)
    result, i = [], 0
    while (prefix + "discriminator_%d.model0.0.weight" % i) in state:
        result.append(nlayer_discriminator(state, prefix + "discriminator_%d." % i, inp, training))
        inp = F.avg_pool2d(inp, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)   # :92-93
        i += 1
    return result


def mask_discriminator(state, vocab, objs, layout_masks, training=True, prefix=""):
    """`MultiscaleMaskDiscriminator2.forward` (discriminator.py:264-308): per real object,
    one-hot(class) broadcast over the mask grid ‖ mask -> NLayerMaskDiscriminator2 at num_D scales."""
    ncls = max(vocab["object_name_to_idx"].values()) + 1
    rows = []
    for b in range(layout_masks.shape[0]):
        m = remove_dummy_objects(objs[b], vocab)
        mk = layout_masks[b][m]
        lab = objs[b][m]
        O, M = mk.shape[0], mk.shape[1]
        one_hot = torch.zeros((O, ncls), dtype=mk.dtype).scatter_(1, lab.view(-1, 1).long(), 1.0)
        rows.append(torch.cat([one_hot.view(O, -1, 1, 1).expand(-1, -1, M, M), mk.unsqueeze(1)], dim=1))
    inp = torch.cat(rows, dim=0).to(state[prefix + "discriminator_0.model0.0.weight"].dtype)     # `.float()` in fp32
    result, i = [], 0
    while (prefix + "discriminator_%d.model0.0.weight" % i) in state:
        result.append(nlayer_discriminator(state, prefix + "discriminator_%d." % i, inp, training))
        inp = F.avg_pool2d(inp, kernel_size=3, stride=2, padding=[1, 1], count_include_pad=False)
        i += 1
    return result


# --------------------------------------------------------------------------- object discriminator
def crop_objects(imgs, objs, boxes, vocab, HH):
    """`crop_bbox_batch_cudnn` + `crop_bbox` (sg2im/bilinear.py:44-94): for every real object, in
    (image, object) order, a bilinear HHxHH crop of its box from its own image; also returns the
    object labels (discriminator.py:253-260)."""
    crops, labels = [], []
    for b in range(imgs.shape[0]):
        m = remove_dummy_objects(objs[b], vocab)
        bx = boxes[b][m]
        n = bx.shape[0]
        if n == 0:
            continue
        pts = torch.stack([bx[:, 0], bx[:, 1], bx[:, 0] + bx[:, 2], bx[:, 1] + bx[:, 3]], dim=1)   # metrics.py:4-8
        pts = 2 * pts - 1                                                                           # bilinear.py:86
        up = torch.linspace(0, 1, steps=HH).view(1, HH)
        down = torch.linspace(1, 0, steps=HH).view(1, HH)
        X = down * pts[:, 0:1] + up * pts[:, 2:3]                                                   # tensor_linspace
        Y = down * pts[:, 1:2] + up * pts[:, 3:4]
        grid = torch.stack([X.view(n, 1, HH).expand(n, HH, HH), Y.view(n, HH, 1).expand(n, HH, HH)], dim=3)
        feats = imgs[b:b + 1].expand(n, -1, -1, -1)
        crops.append(F.grid_sample(feats, grid, mode="bilinear", padding_mode="zeros", align_corners=False))
        labels.append(objs[b][m][:, 0])
    return torch.cat(crops, dim=0), torch.cat(labels, dim=0)


def _batch_norm_affine(state, prefix, x, training):
    """nn.BatchNorm2d (affine, running stats, num_batches_tracked) as build_cnn creates it
    (sg2im/layers.py:132-136)."""
    if training:
        state[prefix + "num_batches_tracked"].add_(1)
    return F.batch_norm(x, state[prefix + "running_mean"], state[prefix + "running_var"], state[prefix + "weight"],
                        state[prefix + "bias"], training, 0.1, 1e-5)


def ac_crop_discriminator(state, vocab, imgs, objs, boxes, crop_size, training=True, prefix=""):
    """`AcCropDiscriminator.forward` -> `AcDiscriminator.forward` (discriminator.py:209-261) with the
    trainer's defaults: arch C4-64-2,C4-128-2,C4-256-2, batch norm, leakyrelu-0.2, 'valid' padding."""
    crops, labels = crop_objects(imgs, objs, boxes, vocab, crop_size)
    p = prefix + "discriminator.cnn.0."
    h = F.conv2d(crops, state[p + "0.weight"], state[p + "0.bias"], stride=2)
    h = F.leaky_relu(_batch_norm_affine(state, p + "1.", h, training), 0.2)
    h = F.conv2d(h, state[p + "3.weight"], state[p + "3.bias"], stride=2)
    h = F.leaky_relu(_batch_norm_affine(state, p + "4.", h, training), 0.2)
    h = F.conv2d(h, state[p + "6.weight"], state[p + "6.bias"], stride=2)
    v = h.reshape(h.shape[0], h.shape[1], -1).mean(dim=2)                                           # GlobalAvgPool
    v = F.linear(v, state[prefix + "discriminator.cnn.2.weight"], state[prefix + "discriminator.cnn.2.bias"])
    real = F.linear(v, state[prefix + "discriminator.real_classifier.weight"],
                    state[prefix + "discriminator.real_classifier.bias"])
    cls = F.linear(v, state[prefix + "discriminator.obj_classifier.weight"],
                   state[prefix + "discriminator.obj_classifier.bias"])
    return real, F.cross_entropy(cls, labels), crops


def bce_loss(x, target):
    """sg2im/losses.py:23-41."""
    return (x.clamp(min=0) - x * target + (1 + (-x.abs()).exp()).log()).mean()


# --------------------------------------------------------------------------- losses
def hinge_loss(x, target_is_real, for_discriminator):
    """`GANLoss.loss`, hinge branch (spade/models/networks/loss.py:65-76)."""
    if for_discriminator:
        z = torch.zeros_like(x)
        return -torch.mean(torch.min(x - 1, z)) if target_is_real else -torch.mean(torch.min(-x - 1, z))
    assert target_is_real
    return -torch.mean(x)


def gan_loss_multiscale(preds, target_is_real, for_discriminator):
    """`GANLoss.__call__` on the list-of-lists D output (loss.py:78-98): mean over scales of the
    loss on each scale's LAST map."""
    return sum(hinge_loss(p[-1], target_is_real, for_discriminator) for p in preds) / len(preds)


VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512)
VGG19_TAPS = (1, 6, 11, 20, 29)     # features[] index of relu1_1 .. relu5_1: the last layer of each slice


def vgg19_features(state, x):
    """`VGG19.forward` (spade/models/networks/architecture.py:93-123): the outputs of the five slices
    features[0:2], [2:7], [7:12], [12:21], [21:30] of torchvision's vgg19 (configuration 'E':
    conv3x3 pad 1 + ReLU, MaxPool2d(2, 2)).  `state` uses the reference module's keys
    (`slice{k}.{features index}.weight|bias`)."""
    bounds = (2, 7, 12, 21, 30)
    outs, idx = [], 0
    for v in VGG19_CFG:
        k = next(i for i, b in enumerate(bounds) if idx < b) + 1
        if v == "M":
            x = F.max_pool2d(x, kernel_size=2, stride=2)
            idx += 1
        else:
            x = F.relu(F.conv2d(x, state["slice%d.%d.weight" % (k, idx)], state["slice%d.%d.bias" % (k, idx)],
                                padding=1))
            idx += 2
            if idx - 1 in VGG19_TAPS:
                outs.append(x)
    return outs


def vgg_loss(state, x, y):
    """`VGGLoss.forward` (spade/models/networks/loss.py:102-117)."""
    weights = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)
    fx, fy = vgg19_features(state, x), vgg19_features(state, y)
    loss = 0
    for w, a, b in zip(weights, fx, fy):
        loss = loss + w * F.l1_loss(a, b.detach())
    return loss


def generator_losses(opt, d_state, batch, model_out, training=True, dobj_state=None, vgg_state=None,
                     dmask_state=None):
    """`Pix2PixModel.compute_generator_loss` (sg2im/pix2pix_model.py:65-143) with mask_size 0; the VGG
    term (:111-113) unless --no_vgg_loss; the object-discriminator terms (:115-121) when
    use_img_disc == 0."""
    imgs, objs, boxes, masks = batch[0], batch[1], batch[2], batch[6]
    imgs_pred, boxes_pred, masks_pred = model_out
    H = opt.image_size[0]
    G = {}
    if not opt.skip_graph_model:                                                              # :71-85
        l = F.smooth_l1_loss(boxes_pred.reshape(-1, 4), boxes.reshape(-1, 4), reduction="none") \
            * opt.bbox_pred_loss_weight
        flat = objs.reshape(-1, objs.shape[-1])
        mask = (flat.sum(1, keepdim=True) != 0) if objs.shape[-1] > 1 else (flat != 0)
        mask = mask.to(torch.float32)
        l = l * mask
        G["bbox_pred_all"] = l.view(boxes.shape).sum(dim=[1, 2]) / mask.view(boxes.shape[0], boxes.shape[1]).sum(dim=1)
        G["bbox_pred"] = G["bbox_pred_all"].mean()
        if masks is not None:                                                                 # :88-92
            M = masks.shape[-1]
            bce = F.binary_cross_entropy(masks_pred.view(-1, M, M), masks.view(-1, M, M).to(masks_pred.dtype),
                                         reduction="none").mean(dim=(1, 2))
            G["masks_pred"] = (bce[mask.bool().nonzero()[:, 0]] * opt.mask_pred_loss_weight).mean()
    if not opt.skip_generation:
        fake = multiscale_discriminator(d_state, opt.vocab, H, imgs_pred, objs, boxes, training,
                                        layout_masks=masks)                                   # :96
        G["GAN_Img"] = gan_loss_multiscale(fake, True, False) * opt.discriminator_img_loss_weight
        if not opt.no_ganFeat_loss:                                                           # :99-109
            real = multiscale_discriminator(d_state, opt.vocab, H, imgs, objs, boxes, training, layout_masks=masks)
            feat = torch.zeros(())
            for i in range(len(fake)):
                for j in range(len(fake[i]) - 1):
                    feat = feat + F.l1_loss(fake[i][j], real[i][j].detach()) * opt.lambda_feat / len(fake)
            G["GAN_Feat"] = feat
        if not opt.no_vgg_loss:                                                               # :111-113
            G["VGG"] = vgg_loss(vgg_state, imgs_pred, imgs) * opt.lambda_vgg
        if not opt.use_img_disc:                                                              # :115-121
            scores_fake, ac_loss, _ = ac_crop_discriminator(dobj_state, opt.vocab, imgs_pred, objs, boxes,
                                                            opt.crop_size, training)
            G["GAN_Obj"] = hinge_loss(scores_fake, True, False) * opt.discriminator_obj_loss_weight
            G["GAN_Ac"] = ac_loss * opt.ac_loss_weight
            if opt.mask_size > 0 and masks_pred is not None:                                  # :124-138
                mfake = mask_discriminator(dmask_state, opt.vocab, objs, masks_pred, training)
                G["GAN_Mask"] = gan_loss_multiscale(mfake, True, False) * opt.discriminator_img_loss_weight
                if not opt.no_ganFeat_loss:
                    mreal = mask_discriminator(dmask_state, opt.vocab, objs, masks, training)
                    feat = torch.zeros(())
                    for i in range(len(mfake)):
                        for j in range(len(mfake[i]) - 1):
                            feat = feat + F.l1_loss(mfake[i][j], mreal[i][j].detach()) * opt.lambda_feat / len(mfake)
                    G["GAN_Mask_Feat"] = feat
    G["total_loss"] = torch.stack([v for k, v in G.items() if k != "bbox_pred_all"]).sum()    # :141-142
    return G


def discriminator_losses(opt, d_state, batch, model_out, training=True, dobj_state=None, dmask_state=None):
    """`Pix2PixModel.compute_discriminator_loss` (pix2pix_model.py:145-202)."""
    imgs, objs, boxes, masks = batch[0], batch[1], batch[2], batch[6]
    imgs_pred = model_out[0].detach()
    H = opt.image_size[0]
    fake = multiscale_discriminator(d_state, opt.vocab, H, imgs_pred, objs, boxes, training, layout_masks=masks)  # :159
    real = multiscale_discriminator(d_state, opt.vocab, H, imgs, objs, boxes, training, layout_masks=masks)       # :161
    D = {"D_img_fake": gan_loss_multiscale(fake, False, True),
         "D_img_real": gan_loss_multiscale(real, True, True)}
    D["total_img_loss"] = D["D_img_fake"] + D["D_img_real"]                                    # :166
    if not opt.use_img_disc:
        with torch.no_grad():                                                                  # :168-172: logged only
            wrong = multiscale_discriminator(d_state, opt.vocab, H, imgs, objs, boxes, training, layout_masks=masks)
            D["D_img_wrong"] = gan_loss_multiscale(wrong, False, True) * (1 / 2) * (.5)
        s_real, ac_real, _ = ac_crop_discriminator(dobj_state, opt.vocab, imgs, objs, boxes, opt.crop_size, training)
        s_fake, ac_fake, _ = ac_crop_discriminator(dobj_state, opt.vocab, imgs_pred, objs, boxes, opt.crop_size,
                                                   training)
        r, f = s_real.reshape(-1), s_fake.reshape(-1)                                          # losses.py:70-87
        D["D_obj"] = (bce_loss(r, torch.ones_like(r)) + bce_loss(f, torch.zeros_like(f))) * 0.5
        D["D_ac_real"], D["D_ac_fake"] = ac_real, ac_fake
        D["total_obj_loss"] = D["D_obj"] + D["D_ac_real"] + D["D_ac_fake"]                     # :185
        if opt.mask_size > 0 and model_out[2] is not None:                                     # :188-196
            mfake = mask_discriminator(dmask_state, opt.vocab, objs, model_out[2].detach(), training)
            mreal = mask_discriminator(dmask_state, opt.vocab, objs, masks, training)
            D["D_mask_fake"] = gan_loss_multiscale(mfake, False, True) * 0.5
            D["D_mask_real"] = gan_loss_multiscale(mreal, True, True) * 0.5
            D["total_mask_loss"] = D["D_mask_fake"] + D["D_mask_real"]
    return D


# --------------------------------------------------------------------------- converse REINFORCE
def calc_log_p(converse_weights, rels, rel_mat):
    """`calc_prob(log=True)` + `calc_log_p` (scripts/graphs_utils.py:109-123): per-sample log-probability
    of the converse edges the data loader drew; rel_mat (B,P,P+1) holds the draw counts."""
    P = converse_weights.shape[0]
    w = torch.cat([converse_weights, torch.zeros(P, 1).to(converse_weights)], dim=-1)
    e = torch.exp(w)
    w_sum = torch.sum(e[:, list(rels) + [P]], dim=1) - torch.diagonal(e)
    log_prob = w - torch.log(w_sum.view(P, 1))
    return torch.sum(log_prob * rel_mat, dim=[1, 2])


def converse_loss(converse_param, vocab, bbox_pred_all, conv_counts):
    """scripts/train.py:343-345,370-378: normalised per-sample box loss x log-probability of the draws."""
    meta = [vocab["pred_name_to_idx"][p] for p in ("__padding__", "__in_image__")]
    non_meta = set(vocab["pred_name_to_idx"].values()) - set(meta)
    eps = float(torch.finfo(torch.float32).eps)
    r = bbox_pred_all.detach()
    if r.shape[0] > 1:
        r = (r - r.mean()) / (r.std() + eps)
    triu = torch.triu(converse_param, diagonal=0)                       # get_conv_converse (sg2im/model.py:10-13)
    return torch.mean(r * calc_log_p(triu + triu.t(), non_meta, conv_counts))


# --------------------------------------------------------------------------- train step
def make_adam_groups(sg_state, g_state, lr):
    """Param groups of scripts/train.py:316-322: everything at `lr`, except
    `trans_candidates_weights` at 1e-2; `converse_candidates_weights` has its own optimizer.
    Parameters = floating tensors that require grad."""
    base, trans, seen = [], [], set()
    trans_id = id(sg_state.get("trans_candidates_weights"))
    for k, v in list(sg_state.items()) + list(g_state.items()):
        if not (torch.is_tensor(v) and v.requires_grad) or id(v) in seen:
            continue
        if k == "converse_candidates_weights":
            continue
        seen.add(id(v))            # the transitive weights are registered under six names (model.py:32,45)
        (trans if id(v) == trans_id else base).append(v)
    return [{"params": base, "lr": lr}, {"params": trans, "lr": 1e-2}]


class TrainState:
    """Leaf tensors + optimizers of one replica (what `scripts.train.main` builds at :312-329)."""

    def __init__(self, opt, sg_state, g_state, d_state, dobj_state=None, vgg_state=None, dmask_state=None,
                 mask_noise=None):
        self.opt, self.sg, self.g, self.d, self.dobj = opt, sg_state, g_state, d_state, dobj_state
        self.vgg = vgg_state                                   # frozen VGG19 weights (None with --no_vgg_loss)
        self.dmask, self.mask_noise = dmask_state, mask_noise  # mask branch (--mask_size > 0)
        if dmask_state is not None:
            pm = [v for v in dmask_state.values() if torch.is_tensor(v) and v.requires_grad]
            self.optimizer_d_mask = torch.optim.Adam(pm, lr=opt.mask_learning_rate, betas=(opt.beta1, 0.999))  # :88-90
        if dobj_state is not None:
            po = [v for v in dobj_state.values() if torch.is_tensor(v) and v.requires_grad]
            self.optimizer_d_obj = torch.optim.Adam(po, lr=opt.learning_rate, betas=(opt.beta1, 0.999))  # :79-81
        self.optimizer = torch.optim.Adam(make_adam_groups(sg_state, g_state, opt.learning_rate))
        if "converse_candidates_weights" in sg_state:                      # scripts/train.py:312-323
            self.optimizer_converse = torch.optim.Adam([{"params": [sg_state["converse_candidates_weights"]], "lr": 1e-2}])
        d_params = [v for v in d_state.values() if torch.is_tensor(v) and v.requires_grad]
        self.optimizer_d_img = torch.optim.Adam(d_params, lr=opt.img_learning_rate,
                                                betas=(opt.beta1, 0.999))          # meta_models.py:67-69


def train_step(ts, batch):
    """One iteration of scripts/train.py:353-393 (+ :468-485) with learned_converse=0 (VGG term when
    `ts.vgg` is given and --no_vgg_loss is absent; object discriminator when use_img_disc=0).  Returns (G_losses, D_losses, imgs_pred)."""
    opt = ts.opt
    imgs, objs, boxes, triplets, _, triplet_type, masks = batch[:7]
    H = opt.image_size[0]
    _, boxes_pred, masks_pred = sg2layout_forward(ts.sg, opt.vocab, objs, triplets, triplet_type,
                                                  mask_noise=ts.mask_noise)                   # meta_models.py:43
    layout_masks = masks_pred if masks is None else masks                                     # :48
    imgs_pred = None
    if not opt.skip_generation:                                                               # meta_models.py:44
        imgs_pred = generator_forward(ts.g, opt.vocab, H, objs, boxes, True,
                                      num_upsampling_layers=opt.num_upsampling_layers,
                                      layout_masks=layout_masks)                              # :47-49 (GT boxes)
    model_out = (imgs_pred, boxes_pred, masks_pred)
    G = generator_losses(opt, ts.d, batch, model_out, dobj_state=ts.dobj, vgg_state=ts.vgg,
                         dmask_state=ts.dmask)                                                # train.py:361
    ts.optimizer.zero_grad()
    for v in list(ts.d.values()) + list((ts.dobj or {}).values()) + list((ts.dmask or {}).values()):
        if torch.is_tensor(v) and v.grad is not None:
            v.grad = None
    G["total_loss"].backward()                                                                # :366-368
    ts.optimizer.step()
    if opt.learned_converse:                                                                  # :370-381
        G["loss_conv"] = converse_loss(ts.sg["converse_candidates_weights"], opt.vocab, G["bbox_pred_all"], batch[4])
        ts.optimizer_converse.zero_grad()
        G["loss_conv"].backward()
        ts.optimizer_converse.step()
        G["loss_conv"] = G["loss_conv"].detach()
    if opt.skip_generation or opt.freeze_options == "generation":                             # train.py:388
        return G, {}, None
    D = discriminator_losses(opt, ts.d, batch, model_out, dobj_state=ts.dobj, dmask_state=ts.dmask)   # :390
    ts.optimizer_d_img.zero_grad()                                                            # :470-472
    D["total_img_loss"].backward()
    ts.optimizer_d_img.step()
    if not opt.use_img_disc:                                                                  # :478-480
        ts.optimizer_d_obj.zero_grad()
        D["total_obj_loss"].backward()
        ts.optimizer_d_obj.step()
    if opt.mask_size > 0 and "total_mask_loss" in D:                                          # :482-485
        ts.optimizer_d_mask.zero_grad()
        D["total_mask_loss"].backward()
        ts.optimizer_d_mask.step()
    return G, D, imgs_pred.detach()
