"""Base class name kept from the reference (spade/models/networks/base_network.py)."""
import torch.nn as nn


class BaseNetwork(nn.Module):
    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print('Network [%s] was created. Total number of parameters: %.1f million.' % (type(self).__name__, n / 1e6))

    def init_weights(self, init_type='normal', gain=0.02):
        # never applied by the trainer's meta models; modules keep PyTorch's default init
        raise NotImplementedError("init_weights is dead code in the reference's training path")
