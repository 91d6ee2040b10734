"""MI355X-native counterparts of the flat unconditional flow of ``models/flow/blocks.py`` (:95-128, :276-319, :531-559,
:692-704): same class names, constructor signatures and state-dict keys.

The modules own the parameters; evaluation happens in ``seq.FlowEngine`` (csrc/seq.hip): one launch per MLP layer with
the scale and translation nets of a coupling batched, one launch for everything between two MLP evaluations, the whole
pass replayed from a hipGraph.  Called where autograd records (grad mode on, trainable parameters: the flow stage of
experiments/behavior_net.py:703-714), the forward direction runs in ``seq_train.FlowTrainEngine`` instead and comes back with
a graph whose backward is the HIP backward pass (csrc/seq_train.hip).
"""
from __future__ import annotations

import torch
from torch import nn

from ...lib.modules import ActNorm, BasicFullyConnectedNet


class Shuffle(nn.Module):
    """models/flow/blocks.py:692-704: a fixed random permutation of the channels (drawn with torch.randperm)."""

    def __init__(self, in_channels, **kwargs):
        super().__init__()
        self.in_channels = in_channels
        idx = torch.randperm(in_channels)
        self.register_buffer("forward_shuffle_idx", idx)
        self.register_buffer("backward_shuffle_idx", torch.argsort(idx))


class DoubleVectorCouplingBlock2(nn.Module):
    """models/flow/blocks.py:276-319 ("support uneven inputs"): two affine half-couplings, scale nets with a tanh head."""

    def __init__(self, in_channels, hidden_dim, hidden_depth=2):
        super().__init__()
        dim1 = (in_channels // 2) + (in_channels % 2)
        dim2 = in_channels // 2
        self.s = nn.ModuleList([BasicFullyConnectedNet(dim=dim1, out_dim=dim2, depth=hidden_depth, hidden_dim=hidden_dim,
                                                       use_tanh=True) for _ in range(2)])
        self.t = nn.ModuleList([BasicFullyConnectedNet(dim=dim1, out_dim=dim2, depth=hidden_depth, hidden_dim=hidden_dim,
                                                       use_tanh=False) for _ in range(2)])


class UnconditionalFlatDoubleCouplingFlowBlock2(nn.Module):
    """models/flow/blocks.py:531-559: ActNorm, coupling, shuffle."""

    def __init__(self, in_channels, hidden_dim, hidden_depth):
        super().__init__()
        self.norm_layer = ActNorm(in_channels, logdet=True)
        self.coupling = DoubleVectorCouplingBlock2(in_channels, hidden_dim, hidden_depth)
        self.shuffle = Shuffle(in_channels)


class UnconditionalFlow2(nn.Module):
    """models/flow/blocks.py:95-128 ("Flat").  ``forward(x)`` -> (z, logdet), ``forward(z, reverse=True)`` -> x; tensors are
    [B, C, 1, 1] (or [B, C]) and come back as [B, C, 1, 1], as the reference's blocks return them."""

    def __init__(self, in_channels, hidden_dim, hidden_depth, n_flows):
        super().__init__()
        self.in_channels = in_channels
        self.mid_channels = hidden_dim
        self.num_blocks = hidden_depth
        self.n_flows = n_flows
        self.sub_layers = nn.ModuleList([UnconditionalFlatDoubleCouplingFlowBlock2(in_channels, hidden_dim, hidden_depth)
                                         for _ in range(n_flows)])
        self._engine = None
        self._train_engine = None

    def engine(self):
        if self._engine is None:
            from ... import seq
            object.__setattr__(self, "_engine", seq.FlowEngine(self))
        return self._engine

    def train_engine(self, **adam):
        """The training plans of this flow (``seq_train.FlowTrainEngine``); ``adam``: the hyper-parameters of its fused
        optimiser step (used by ``experiments.behavior_net``; autograd callers bring their own optimiser)."""
        if self._train_engine is None:
            from ... import seq_train
            object.__setattr__(self, "_train_engine", seq_train.FlowTrainEngine(self, **adam))
        return self._train_engine

    def forward(self, x, reverse=False):
        if reverse:
            return self.engine().reverse(x)[:, :, None, None]
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ... import seq_train
            out, logdet = seq_train.flow_autograd(self.train_engine(), x)
            return out[:, :, None, None], logdet
        out, logdet = self.engine().forward(x)
        return out[:, :, None, None], logdet

    def reverse(self, out):
        return self(out, reverse=True)
