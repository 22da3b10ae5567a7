"""fp64 evaluations of the oracle (test infrastructure, like the rest of this package).

The fp32 oracle is one fp32 evaluation of the reference's algorithm; the HIP path is another.  Where two fp32
evaluations are expected to differ by more than a tolerance allows (long reductions, saturating tanh, gate decisions),
the yardstick is the SAME functional code run in double precision: `state_to64` / `trainstate_to64` / `batch_to64`
turn an oracle state into fp64 leaves, `generated_image64` is the forward half of `oracle.train_step`
(scripts/train.py:353-358 -> sg2im/meta_models.py:43-49) in fp64."""
import torch

from . import functional as OF


def state_to64(state, memo=None):
    """fp64 leaf copy of an oracle state dict; aliased tensors (the six names of the transitive weights) stay aliased."""
    if state is None:
        return None
    memo = {} if memo is None else memo
    out = {}
    for k, v in state.items():
        if torch.is_tensor(v) and v.is_floating_point():
            if id(v) not in memo:
                memo[id(v)] = v.detach().double().clone().requires_grad_(v.requires_grad)
            out[k] = memo[id(v)]
        else:
            out[k] = v
    return out


def trainstate_to64(ts, oracle_mod=None):
    noise = None if ts.mask_noise is None else ts.mask_noise.double()
    return OF.TrainState(ts.opt, state_to64(ts.sg), state_to64(ts.g), state_to64(ts.d), state_to64(ts.dobj),
                         state_to64(ts.vgg), state_to64(ts.dmask), noise)


def batch_to64(batch):
    return tuple(t.double() if (torch.is_tensor(t) and t.is_floating_point()) else t for t in batch)


def generated_image64(ts, batch):
    """imgs_pred of `oracle.train_step(ts, batch)` evaluated in fp64 (forward only, training-mode normalisation and
    spectral-norm iteration as in the step).  `ts` is not modified: the fp64 copy takes the buffer updates."""
    ts64 = trainstate_to64(ts)
    opt = ts64.opt
    b64 = batch_to64(batch)
    with torch.no_grad():
        _, _, masks_pred = OF.sg2layout_forward(ts64.sg, opt.vocab, b64[1], b64[3], b64[5], mask_noise=ts64.mask_noise)
        return OF.generator_forward(ts64.g, opt.vocab, opt.image_size[0], b64[1], b64[2], True,
                                    num_upsampling_layers=opt.num_upsampling_layers,
                                    layout_masks=masks_pred if b64[6] is None else b64[6]).detach()
