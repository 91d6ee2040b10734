"""``UnsupervisedTransformer2`` (models/flow/simple_flow.py:136-176): the flow over behaviour codes that BASELINE config 5
samples from (experiments/behavior_net.py:1086-1100, :1173-1177).  Same constructor keywords and state-dict keys."""
from __future__ import annotations

import torch
from torch import nn

from .blocks import UnconditionalFlow2


class UnsupervisedTransformer2(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        in_channels = kwargs["flow_in_channels"]
        self.in_channels = in_channels
        self.flow = UnconditionalFlow2(in_channels=in_channels, hidden_dim=kwargs["flow_mid_channels"],
                                       hidden_depth=kwargs["flow_hidden_depth"], n_flows=kwargs["n_flows"])

    def sample(self, shape, device="cuda"):
        """models/flow/simple_flow.py:154-158 (the reference defaults to the CPU; this path has none)."""
        z_tilde = torch.randn(shape, device=device)
        return self.reverse(z_tilde).squeeze(dim=-1).squeeze(dim=-1)

    def forward(self, input, reverse=False, train=False):
        if reverse:
            return self.reverse(input)
        if train:
            raise NotImplementedError("train=True returns every block's output and running logdet (models/flow/blocks.py:116-119); "
                                      "no call site of the reference passes it (experiments/behavior_net.py:705 trains the flow "
                                      "through forward(input)): the fused blocks here do not keep the per-block values")
        return self.flow(input)

    def reverse(self, out):
        return self.flow(out, reverse=True)

    def get_last_layer(self):
        return getattr(self.flow.sub_layers[-1].coupling.t[-1].main[-1], "weight")
