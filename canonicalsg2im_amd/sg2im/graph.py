"""Scene-graph convolution (reference: sg2im/graph.py).

The reference loops over samples in Python (graph.py:63-64, 85-107).  Here one CSR of incident
triplets per image is built on the device and three kernels do the work for the whole batch:
gather-concat (K2), the two MLPs on the fp32-MFMA GEMM with fused ReLU (K3), and the
confidence-weighted segment average (K4+K5) — no atomics, fixed summation order."""
import torch
import torch.nn as nn

from .. import ops
from .layers import Linear, _FusedActivation, build_mlp

ORIGINAL_EDGE, TRANSITIVE_EDGE, SYMMETRIC_EDGE, ANTI_SYMMETRIC_EDGE = 0, 1, 2, 3   # sg2im/data/base_dataset.py:7-10


def _init_weights(module):
    if isinstance(module, nn.Linear):
        nn.init.kaiming_normal_(module.weight)


class GraphTopology:
    """Per-batch index structures shared by all gconv layers of one forward."""

    def __init__(self, triplets, pred_indicators, num_objs):
        self.triplets = triplets.contiguous()
        self.valid = pred_indicators.to(torch.uint8).contiguous()
        self.row_ptr, self.col = ops.graph_csr(self.triplets, num_objs)


class GraphTripleConv(nn.Module):
    """A single layer of scene graph convolution (same constructor as the reference)."""

    def __init__(self, obj_input_dim, object_output_dim, predicate_input_dim, predicate_output_dim, hidden_dim,
                 num_attributes, pooling='avg', mlp_normalization='none', predicates_transitive_weights=None,
                 return_new_p_vecs=True):
        super().__init__()
        assert pooling in ['sum', 'avg'], 'Invalid pooling "%s"' % pooling
        self.return_new_p_vecs = return_new_p_vecs
        self.hidden_dim = hidden_dim
        self.num_attributes = num_attributes
        self.predicate_output_dim = predicate_output_dim
        self.pooling = pooling          # asserted only: the layer always averages (graph.py:31,101-106)
        self.net1 = build_mlp([2 * obj_input_dim + predicate_input_dim, hidden_dim,
                               2 * hidden_dim + predicate_output_dim], batch_norm=mlp_normalization)
        self.net1.apply(_init_weights)
        self.net2 = build_mlp([hidden_dim, hidden_dim, object_output_dim], batch_norm=mlp_normalization)
        self.net2.apply(_init_weights)
        self.predicates_transitive_weights = predicates_transitive_weights

    def forward(self, obj_vecs, pred_vecs, edges, pred_indicators, triplet_type, predicate_ids, topology=None):
        if topology is None:
            triplets = torch.stack([edges[..., 0], predicate_ids, edges[..., 1]], dim=-1)
            topology = GraphTopology(triplets, pred_indicators, obj_vecs.size(1))
        t = topology
        cur_t = ops.gather_concat(obj_vecs, pred_vecs, t.triplets, t.row_ptr, t.col)      # graph.py:63-66
        # net1 (:67).  Its final ReLU feeds ONLY the segment average below, whose backward then returns the gradient of
        # the pre-activation (h_is_relu): no activation-derivative pass over the (B, T, 2H + Dp) tensor
        last = self.net1[-2] if len(self.net1) >= 2 else None
        relu_tail = (isinstance(last, Linear) and last.fused_slope == 0.0 and isinstance(self.net1[-1], _FusedActivation)
                     and last.out_features % 4 == 0)
        if relu_tail:
            h = cur_t
            for m in list(self.net1)[:-2]:
                h = m(h)
            h = last(h, grad_is_pre=True)
        else:
            h = self.net1(cur_t)
        # confidence gate (:69-74): P-sized sigmoid stays a torch op so autograd reaches the weights
        tt = triplet_type
        sig = torch.sigmoid(self.predicates_transitive_weights)
        conf = (tt == ORIGINAL_EDGE).to(h.dtype) + (tt == TRANSITIVE_EDGE).to(h.dtype) * sig[predicate_ids]
        pooled, new_p = ops.segment_avg(h, conf, t.valid, t.triplets, t.row_ptr, t.col, self.hidden_dim,
                                        self.predicate_output_dim, h_is_relu=relu_tail)   # :76-109
        new_obj = self.net2(pooled)                                                       # :110
        if not self.return_new_p_vecs:
            new_p = pred_vecs
        return new_obj, new_p


def get_predicates_weights(num_preds, learned_init):
    if learned_init == 'uniform':
        w = nn.Parameter(torch.zeros(num_preds), requires_grad=True)
        w.data.uniform_(-1, 1)
    elif learned_init in ('-4', '0', '4'):
        w = nn.Parameter(float(learned_init) * torch.ones(num_preds), requires_grad=True)
    else:
        raise ValueError()
    return w
