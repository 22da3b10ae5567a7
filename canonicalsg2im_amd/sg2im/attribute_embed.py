"""One embedding table per attribute (`att_emb_0`, `att_emb_1`, ...: the reference's names,
sg2im/attribute_embed.py:18-48) looked up and concatenated by ONE gather kernel; an optional Linear over
the concatenation (`attribute_fc_gen`) exists iff there are several attributes or the caller asks for it
(the discriminator does, discriminator.py:71-72)."""
import torch.nn as nn

from .. import ops
from .layers import Linear


class AttributeEmbeddings(nn.Module):
    def __init__(self, attributes, embedding_dim, use_attr_fc_gen=False):
        super().__init__()
        self.num_attributes = len(attributes)
        width = self.num_attributes * embedding_dim
        if self.num_attributes > 1 or use_attr_fc_gen:
            self.attribute_fc_gen = Linear(width, width)
        for k, values in enumerate(attributes.values()):
            self.add_module("att_emb_%d" % k, nn.Embedding(max(values.values()) + 1, embedding_dim))

    def forward(self, x):
        """ids (B, O, A) int64 -> vectors (B, O, A*E)."""
        out = ops.embed(x, [getattr(self, "att_emb_%d" % k).weight for k in range(x.size(-1))])
        fc = getattr(self, 'attribute_fc_gen', None)
        return out if fc is None else fc(out)
