"""Scene graph -> per-object vectors, boxes and (optionally) masks: the reference's `Sg2LayoutModel`
surface (sg2im/model.py:15-124) — constructor from `opt`, attributes `vocab, gconvs, box_net, mask_net,
attribute_embedding, pred_embeddings, trans_candidates_weights, converse_candidates_weights`,
`forward(objs, triplets, triplet_type, boxes_gt=None, masks_gt=None)` and the same state_dict keys.

Execution differs: the per-image CSR of the graph is built once per forward and shared by all
GraphTripleConv layers; gather, segment average and the MLPs run on the HIP kernels."""
import torch
import torch.nn as nn

from .. import ops
from .attribute_embed import AttributeEmbeddings
from .graph import GraphTopology, GraphTripleConv, get_predicates_weights
from .layers import BatchNormAct, Conv2d, Interpolate, _FusedActivation, build_mlp


def get_conv_converse(model):
    """Symmetrised upper triangle of the converse-candidate weights, from a model or a checkpoint dict
    (reference model.py:10-13)."""
    key = "sg_to_layout.module.converse_candidates_weights"
    w = model[key] if isinstance(model, dict) else model.sg_to_layout.module.converse_candidates_weights
    upper = torch.triu(w, diagonal=0)
    return upper + upper.t()


class Sg2LayoutModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        cfg = vars(opt)
        self.args, self.vocab = cfg, cfg["vocab"]
        self.image_size = cfg["image_size"]
        self.layout_noise_dim, self.mask_noise_dim = cfg["layout_noise_dim"], cfg.get("mask_noise_dim")
        E, D, hidden = cfg["embedding_dim"], cfg["gconv_dim"], cfg["gconv_hidden_dim"]
        n_attr = len(self.vocab['attributes'])
        n_pred = len(self.vocab['pred_idx_to_name'])

        self.attribute_embedding = AttributeEmbeddings(self.vocab['attributes'], E)
        self.pred_embeddings = nn.Embedding(n_pred, E)
        # ONE Parameter for the transitive-edge confidences, registered here and again inside every layer
        # (six state_dict names, reference model.py:32,45)
        self.trans_candidates_weights = get_predicates_weights(n_pred, opt.learned_init)
        self.converse_candidates_weights = get_predicates_weights((n_pred, n_pred), opt.learned_init)
        self.gconvs = nn.ModuleList(
            GraphTripleConv(obj_input_dim=(n_attr * E if layer == 0 else D), predicate_input_dim=(E if layer == 0 else D),
                            object_output_dim=D, predicate_output_dim=D, hidden_dim=hidden, num_attributes=n_attr,
                            mlp_normalization=cfg["mlp_normalization"], pooling=cfg["gconv_pooling"],
                            predicates_transitive_weights=self.trans_candidates_weights)
            for layer in range(cfg["gconv_num_layers"]))
        self.box_net = build_mlp([D, hidden, 4], batch_norm=cfg["mlp_normalization"], final_nonlinearity=None)

        self.mask_noise = None       # tests pin the (1, mask_noise_dim) noise row here; None = draw per forward
        self.mask_net = None
        if (cfg["mask_size"] or 0) > 0:
            self.mask_net = self._build_mask_net(cfg['g_mask_dim'], cfg["mask_size"])

    def _build_mask_net(self, dim, mask_size):
        """log2(M) stages of [nearest x2, Conv3x3, BatchNorm2d, ReLU] and a final Conv1x1 -> 1 channel
        (reference model.py:67-79) with the same nn.Sequential indices, hence the same state_dict keys;
        BatchNorm and ReLU are one fused pass, the ReLU slot keeps a placeholder."""
        stages = mask_size.bit_length() - 1
        if mask_size < 1 or (1 << stages) != mask_size:
            raise ValueError('Mask size must be a power of 2')
        mods = []
        for _ in range(stages):
            mods += [Interpolate(scale_factor=2, mode='nearest'), Conv2d(dim, dim, kernel_size=3, padding=1),
                     BatchNormAct(dim, fused_slope=0.0), _FusedActivation()]
        mods.append(Conv2d(dim, 1, kernel_size=1))
        return nn.Sequential(*mods)

    def create_mask_vecs(self, objs, obj_vecs):
        """Every object's vector followed by ONE noise row shared by the whole batch (reference model.py:81-88)."""
        B, O = objs.shape[0], objs.shape[1]
        row = self.mask_noise
        if row is None:
            row = torch.randn((1, self.mask_noise_dim), dtype=obj_vecs.dtype, device=obj_vecs.device)
        return torch.cat([obj_vecs, row.to(obj_vecs.device).expand(B, O, self.mask_noise_dim)], dim=-1)

    def attribute_embedding_pred(self, p):
        return ops.embed(p.unsqueeze(-1), [self.pred_embeddings.weight])

    def forward(self, objs, triplets, triplet_type, boxes_gt=None, masks_gt=None):
        p = triplets[..., 1]
        edges = triplets[..., 0::2]                                           # (s, o) columns
        is_edge = p != self.vocab["pred_name_to_idx"]["__padding__"]
        topology = GraphTopology(triplets, is_edge, objs.size(1))
        obj_vecs, pred_vecs = self.attribute_embedding(objs), self.attribute_embedding_pred(p)
        for layer in self.gconvs:
            obj_vecs, pred_vecs = layer(obj_vecs, pred_vecs, edges, is_edge, triplet_type, p, topology=topology)
        boxes_pred = self.box_net(obj_vecs)
        masks_pred = None
        if self.mask_net is not None:                                         # reference model.py:118-123
            B, O = objs.shape[0], objs.shape[1]
            scores = self.mask_net(self.create_mask_vecs(objs, obj_vecs).reshape(B * O, -1, 1, 1))
            masks_pred = scores.reshape(B, O, scores.size(2), scores.size(3)).sigmoid()
        return obj_vecs, boxes_pred, masks_pred
