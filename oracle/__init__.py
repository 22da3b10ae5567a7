"""CPU oracle for the CanonicalSg2Im training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it; nothing under `canonicalsg2im_amd/` does.  It is the checker the HIP path is
compared with, never the thing measured or shipped.

What it is: a plain-PyTorch (CPU, fp32) *functional* restatement of the
reference algorithm.  Every function takes a flat `state` dict of tensors whose
keys are the reference's own `state_dict` keys and cites the
reference file:line it follows.  The arithmetic of the reference lives in a
third-party dependency, PyTorch (pinned `torch==0.4.0` in the reference's
`requirements.txt:21`; the README mentions 1.1.0); the reference has no tests or
golden vectors of its own for this path.

Parity pinning: the oracle is pinned against outputs of the reference itself,
imported in the build container from `/root/reference` (torch 2.10 CPU, hence
`grid_sample(align_corners=False)` — SURVEY.md §8c) by
`tests/golden/make_golden.py`; the resulting vectors are committed under
`tests/golden/*.npz` and checked by `tests/test_oracle_golden.py`.
"""
from .functional import *  # noqa: F401,F403
