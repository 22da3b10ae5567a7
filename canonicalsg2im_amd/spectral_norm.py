"""`spectral_norm(module)` with the reference's semantics (torch.nn.utils.spectral_norm: parameters
`weight_orig`, buffers `weight_u` / `weight_v`, one power iteration per training-mode forward call,
state_dict hooks) whose per-call arithmetic runs on the HIP kernels of csrc/spectral.hip.

Registration is torch's own (`SpectralNorm.apply`), so checkpoints, `remove_spectral_norm` and the
state-dict version hooks behave exactly as in the reference; only `compute_weight` is replaced."""
from torch.nn.utils.spectral_norm import SpectralNorm
from torch.nn.utils.spectral_norm import spectral_norm as _torch_spectral_norm

from . import ops


class HipSpectralNorm(SpectralNorm):
    def compute_weight(self, module, do_power_iteration):
        weight = getattr(module, self.name + "_orig")
        if not weight.is_cuda:
            raise RuntimeError("canonicalsg2im_amd ops need HIP (cuda) tensors; got a %s tensor — there is no CPU path"
                               % weight.device)
        if self.dim != 0 or self.n_power_iterations != 1:
            raise NotImplementedError("spectral_norm: only dim=0, n_power_iterations=1 (the reference's settings, "
                                      "architecture.py:35-39) are on the hot path")
        u = getattr(module, self.name + "_u")
        v = getattr(module, self.name + "_v")
        return ops.spectral_weight(weight, u, v, do_power_iteration, self.eps)


def spectral_norm(module, name='weight', n_power_iterations=1, eps=1e-12, dim=None):
    w = getattr(module, name)
    w.data = w.data.contiguous()                 # rows of W.view(Cout, -1) in the reference's (Cin, KH, KW) order
    module = _torch_spectral_norm(module, name, n_power_iterations, eps, dim)
    for hook in module._forward_pre_hooks.values():
        if isinstance(hook, SpectralNorm) and hook.name == name:
            hook.__class__ = HipSpectralNorm
    return module
