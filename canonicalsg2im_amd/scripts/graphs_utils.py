"""Trainer-side pieces of the reference's scripts/graphs_utils.py: the log-probability of the converse
edges the data loader sampled (REINFORCE signal of `--learned_converse`, scripts/train.py:370-381).
(P, P+1) tensors — scalar-sized torch arithmetic; the graph construction itself lives in
sg2im/data/base_dataset.py (device side) and the sampling stays in the host data loader."""
import torch


def calc_prob(converse_weights, rels, log=False):
    """Row-wise softmax of the converse weights over the candidate relations `rels` plus the 'sample
    nothing' column (weight 0), each row excluding its own relation (graphs_utils.py:109-118)."""
    P = converse_weights.shape[0]
    w = torch.cat([converse_weights, converse_weights.new_zeros(P, 1)], dim=-1)          # (P, P+1)
    e = torch.exp(w)
    denom = e[:, list(rels) + [P]].sum(dim=1) - torch.diagonal(e)
    log_prob = w - torch.log(denom).view(P, 1)
    return log_prob if log else torch.exp(log_prob)


def calc_log_p(converse_weights, rels, rel_mat):
    """log-probability of each sample's recorded draws: rel_mat (B, P, P+1) counts (graphs_utils.py:121-123)."""
    return torch.sum(calc_prob(converse_weights, rels, log=True) * rel_mat, dim=[1, 2])
