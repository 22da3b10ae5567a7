"""Per-attribute embedding tables (reference: sg2im/attribute_embed.py:18-48)."""
import torch.nn as nn

from .. import ops
from .layers import Linear


class AttributeEmbeddings(nn.Module):
    def __init__(self, attributes, embedding_dim, use_attr_fc_gen=False):
        super().__init__()
        names = list(attributes)
        if len(names) > 1 or use_attr_fc_gen:
            self.attribute_fc_gen = Linear(len(names) * embedding_dim, len(names) * embedding_dim)
        for i, name in enumerate(names):
            self.add_module("att_emb_%d" % i, nn.Embedding(max(attributes[name].values()) + 1, embedding_dim))
        self.num_attributes = len(names)

    def forward(self, x):
        """x int64 [B, O, A] -> [B, O, A*E]: one fused gather for all attribute columns."""
        tables = [self._modules["att_emb_%d" % k].weight for k in range(x.size(-1))]
        vecs = ops.embed(x, tables)
        if hasattr(self, 'attribute_fc_gen'):
            vecs = self.attribute_fc_gen(vecs)
        return vecs
