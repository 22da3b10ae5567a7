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
    # The oracle is torch on the CPU, and torch's CPU convolutions stop scaling (then regress) well before a 2 x 64-core host
    # is full: on the GPU box one full-width oracle step takes 53 s on 128 threads and 11-13 s on 32 (bench.py's cpu_baseline
    # measures both).  The full-width parity tests are oracle-bound, so the suite runs them on at most 32 threads.
    torch.set_num_threads(min(32, torch.get_num_threads()))


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
