"""Device-side pieces of the reference's data pipeline (sg2im/data/*): only the canonical graph
construction of the packed datasets — the O(O^3) numpy/python step that feeds the hot path."""
from .base_dataset import (ANTI_SYMMETRIC_EDGE, ORIGINAL_EDGE, SYMMETRIC_EDGE, TRANSITIVE_EDGE,  # noqa: F401
                           augmented_relations, canonical_triplets, meta_relations, register_augmented_relations)
