"""Scene graph -> layout model (reference: sg2im/model.py)."""
import torch
import torch.nn as nn

from .attribute_embed import AttributeEmbeddings
from .graph import GraphTopology, GraphTripleConv, get_predicates_weights
from .layers import build_mlp


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
        if args["mask_size"] is not None and args["mask_size"] > 0:
            raise NotImplementedError("mask_size > 0 (mask net) is outside the hot path (SURVEY.md §8f row 4)")

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
        return obj_vecs, boxes_pred, None

    def attribute_embedding_pred(self, p):
        from .. import ops
        return ops.embed(p.unsqueeze(-1), [self.pred_embeddings.weight])
