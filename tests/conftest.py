import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (meta dict, {key: torch tensor})"""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "__meta__"}


def sub_state(arrays, prefix, requires_grad=True, device=None):
    """Strip `prefix` from the keys; float tensors become leaf parameters."""
    out = {}
    for k, v in arrays.items():
        if not k.startswith(prefix):
            continue
        t = v.clone()
        if device is not None:
            t = t.to(device)
        key = k[len(prefix):]
        is_buffer = any(s in key for s in ("running_", "weight_u", "weight_v", "num_batches_tracked"))
        if requires_grad and t.is_floating_point() and not is_buffer:
            t.requires_grad_(True)
        out[key] = t
    return out


def state_from_shapes(shapes, seed, requires_grad=True, device=None):
    """Rebuild the `deterministic_state` a golden file was generated with from its recorded shapes."""
    from canonicalsg2im_amd.synth import deterministic_state
    protos = {k: torch.empty(sh, dtype=getattr(torch, dt)) for k, (sh, dt) in shapes.items()}
    sd = deterministic_state(protos, seed=seed)
    return sub_state(sd, "", requires_grad=requires_grad, device=device)


def assert_close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = a.detach().cpu() if torch.is_tensor(a) else torch.as_tensor(a)
    b = b.detach().cpu() if torch.is_tensor(b) else torch.as_tensor(b)
    assert a.shape == b.shape, "%s shape %s vs %s" % (msg, tuple(a.shape), tuple(b.shape))
    if not torch.allclose(a.double(), b.double(), rtol=rtol, atol=atol):
        d = (a.double() - b.double()).abs()
        rel = d / b.double().abs().clamp_min(1e-12)
        raise AssertionError("%s max abs diff %.3e (max rel %.3e) rtol %g atol %g" %
                             (msg, d.max().item(), rel.max().item(), rtol, atol))


@pytest.fixture(scope="session")
def golden():
    return load_golden
