"""Scene graph -> layout model (reference: sg2im/model.py)."""
import torch
import torch.nn as nn

from .attribute_embed import AttributeEmbeddings
from .graph import GraphTopology, GraphTripleConv, get_predicates_weights
from .layers import BatchNormAct, Conv2d, Interpolate, _FusedActivation, build_mlp


def get_conv_converse(model):
    if isinstance(model, dict):
        base = model["sg_to_layout.module.converse_candidates_weights"]
    else:
        base = model.sg_to_layout.module.converse_candidates_weights
    triu = torch.triu(base, diagonal=0)
    return triu + triu.t()


class Sg2LayoutModel(nn.Module):
    def __init__(self, opt):
        super().__init__()
        args = vars(opt)
        self.args = args
        self.vocab = args["vocab"]
        self.image_size = args["image_size"]
        self.layout_noise_dim = args["layout_noise_dim"]
        self.mask_noise_dim = args.get("mask_noise_dim")
        E = args["embedding_dim"]
        self.attribute_embedding = AttributeEmbeddings(self.vocab['attributes'], E)
        num_preds = len(self.vocab['pred_idx_to_name'])
        self.pred_embeddings = nn.Embedding(num_preds, E)
        num_attributes = len(self.vocab['attributes'].keys())
        self.trans_candidates_weights = get_predicates_weights(num_preds, opt.learned_init)
        self.converse_candidates_weights = get_predicates_weights((num_preds, num_preds), opt.learned_init)
        common = dict(object_output_dim=args["gconv_dim"], predicate_output_dim=args["gconv_dim"],
                      hidden_dim=args["gconv_hidden_dim"], num_attributes=num_attributes,
                      mlp_normalization=args["mlp_normalization"], pooling=args["gconv_pooling"],
                      predicates_transitive_weights=self.trans_candidates_weights)   # ONE parameter, shared
        self.gconvs = nn.ModuleList()
        for i in range(args["gconv_num_layers"]):
            din = num_attributes * E if i == 0 else args["gconv_dim"]
            dp = E if i == 0 else args["gconv_dim"]
            self.gconvs.append(GraphTripleConv(obj_input_dim=din, predicate_input_dim=dp, **common))
        self.box_net = build_mlp([args["gconv_dim"], args["gconv_hidden_dim"], 4],
                                 batch_norm=args["mlp_normalization"], final_nonlinearity=None)
        self.mask_net = None
        self.mask_noise = None       # tests pin the (1, mask_noise_dim) noise row here; None = draw per forward
        if args["mask_size"] is not None and args["mask_size"] > 0:
            self.mask_net = self._build_mask_net(args['g_mask_dim'], args["mask_size"])

    def _build_mask_net(self, dim, mask_size):
        """[Upsample x2, Conv3x3, BatchNorm2d, ReLU] * log2(M) + Conv1x1 -> 1 (reference model.py:67-79);
        same nn.Sequential indices / state_dict keys, BatchNorm + ReLU fused into one pass."""
        layers, cur_size = [], 1
        while cur_size < mask_size:
            layers.append(Interpolate(scale_factor=2, mode='nearest'))
            layers.append(Conv2d(dim, dim, kernel_size=3, padding=1))
            layers.append(BatchNormAct(dim, fused_slope=0.0))
            layers.append(_FusedActivation())
            cur_size *= 2
        if cur_size != mask_size:
            raise ValueError('Mask size must be a power of 2')
        layers.append(Conv2d(dim, 1, kernel_size=1))
        return nn.Sequential(*layers)

    def create_mask_vecs(self, objs, obj_vecs):
        """obj_vecs || one noise row shared by every object of the batch (reference model.py:81-88)."""
        B, O = objs.size(0), objs.size(1)
        noise = self.mask_noise
        if noise is None:
            noise = torch.randn((1, self.mask_noise_dim), dtype=obj_vecs.dtype, device=obj_vecs.device)
        noise = noise.to(obj_vecs.device).repeat((B, O, 1)).view(B, O, self.mask_noise_dim)
        return torch.cat([obj_vecs, noise], dim=-1)

    def forward(self, objs, triplets, triplet_type, boxes_gt=None, masks_gt=None):
        s, p, o = triplets[..., 0], triplets[..., 1], triplets[..., 2]
        edges = torch.stack([s, o], dim=-1)
        pred_indicators = p != self.vocab["pred_name_to_idx"]["__padding__"]
        obj_vecs = self.attribute_embedding(objs)
        pred_vecs = self.attribute_embedding_pred(p)
        topo = GraphTopology(triplets, pred_indicators, objs.size(1))
        for conv in self.gconvs:
            obj_vecs, pred_vecs = conv(obj_vecs, pred_vecs, edges, pred_indicators, triplet_type, p, topology=topo)
        boxes_pred = self.box_net(obj_vecs)
        masks_pred = None
        if self.args["mask_size"] > 0:                                   # model.py:118-123
            B, O = objs.size(0), objs.size(1)
            mask_vecs = self.create_mask_vecs(objs, obj_vecs)
            scores = self.mask_net(mask_vecs.view(B * O, -1, 1, 1))
            masks_pred = scores.reshape(B, O, scores.size(2), scores.size(3)).sigmoid()
        return obj_vecs, boxes_pred, masks_pred

    def attribute_embedding_pred(self, p):
        from .. import ops
        return ops.embed(p.unsqueeze(-1), [self.pred_embeddings.weight])
