"""MI355X-native counterparts of ``ResidualBehaviorNet`` and its parts (models/pose_behavior_rnn.py:125-209, :463-534,
:538-626): the behaviour encoder (one-layer LSTM + the two 1x1 ``NormConv2d`` bottleneck heads) and the residual LSTM
decoder whose roll-out produces the pose sequence BASELINE config 5 renders.

Same class names, constructor signatures, state-dict keys and return values; the recurrences run in
``seq.BehaviorEngine`` (csrc/seq.hip): two launches per time step, the roll-out replayed from a hipGraph.
"""
from __future__ import annotations

import math

import torch
from torch import nn

from ..lib.modules import Linear, NormConv2d


class _LSTMParams(nn.Module):
    """Parameter holder with the keys of ``nn.LSTMCell`` (suffix "") or a one-layer ``nn.LSTM`` (suffix "_l0")."""

    def __init__(self, n_in, n_hidden, suffix=""):
        super().__init__()
        self.input_size, self.hidden_size = n_in, n_hidden
        k = 1.0 / math.sqrt(n_hidden)
        for name, shape in (("weight_ih", (4 * n_hidden, n_in)), ("weight_hh", (4 * n_hidden, n_hidden)),
                            ("bias_ih", (4 * n_hidden,)), ("bias_hh", (4 * n_hidden,))):
            setattr(self, name + suffix, nn.Parameter(torch.empty(*shape).uniform_(-k, k)))


class BEncoder(nn.Module):
    """models/pose_behavior_rnn.py:125-209."""

    def __init__(self, n_in, n_layers, dim_hidden, use_linear, dim_linear, ib=False):
        super().__init__()
        if n_layers != 1:
            raise NotImplementedError("ResidualBehaviorNet builds its encoder with one LSTM layer (:558-565)")
        self.rnn = _LSTMParams(n_in, dim_hidden, "_l0")
        self.n_layer, self.dim_hidden, self.ib = n_layers, dim_hidden, ib
        self.use_linear = use_linear
        self.linear = Linear(dim_hidden, dim_linear) if use_linear else None   # (never applied by the reference either, :178-181)
        if self.ib:
            self.mu_fn = NormConv2d(dim_hidden, dim_hidden, 1)
            self.std_fn = NormConv2d(dim_hidden, dim_hidden, 1)


class ResidualRNNDecoder(nn.Module):
    """models/pose_behavior_rnn.py:463-534."""

    def __init__(self, n_in_out, n_hidden, rnn_type="lstm", use_nin=False):
        super().__init__()
        if rnn_type != "lstm":
            raise NotImplementedError("the reference's GRU decoder never defines n_out (:473-478) and cannot run; LSTM only")
        self.n_in_out, self.n_hidden, self.rnn_type, self.use_nin = n_in_out, n_hidden, rnn_type, use_nin
        self.rnn = _LSTMParams(n_in_out, n_hidden)
        self.n_out = Linear(n_hidden, n_in_out)
        if use_nin:
            self.n_in = Linear(n_in_out, n_in_out)


class ResidualBehaviorNet(nn.Module):
    """models/pose_behavior_rnn.py:538-626."""

    def __init__(self, n_kps, **kwargs):
        super().__init__()
        self.dec_type = kwargs.get("decoder_arch", "lstm")
        self.use_nin_dec = kwargs.get("linear_in_decoder", False)
        self.ib = kwargs.get("information_bottleneck", False)
        self.dim_hidden_b = kwargs["dim_hidden_b"]
        self.b_enc = BEncoder(n_kps, 1, self.dim_hidden_b, use_linear=False, dim_linear=1, ib=self.ib)
        self.decoder = ResidualRNNDecoder(n_in_out=n_kps, n_hidden=self.dim_hidden_b, rnn_type=self.dec_type,
                                          use_nin=self.use_nin_dec)
        self._engine = None
        self._train_engine = None

    def engine(self):
        if self._engine is None:
            from .. import seq
            object.__setattr__(self, "_engine", seq.BehaviorEngine(self))
        return self._engine

    def train_engine(self):
        """The training plans of this net (``seq_train.BehaviorTrainEngine``: the forward pass with everything kept and
        back-propagation through time, csrc/seq_bptt.hip)."""
        if self._train_engine is None:
            from .. import seq_train
            object.__setattr__(self, "_train_engine", seq_train.BehaviorTrainEngine(self))
        return self._train_engine

    def forward(self, x1, x2, len, start_frame=0, sample=False, eps=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # autograd is recording (the cVAE stage of experiments/behavior_net.py:591-660): the same forward pass with every
            # step kept, as one node whose backward is the HIP back-propagation through time
            from .. import seq_train
            if eps is None:
                eps = torch.randn(x1.shape[0], self.dim_hidden_b, device=x1.device, dtype=x1.dtype)
            xs, cs, b, mu, logstd, pre = seq_train.behavior_autograd(self.train_engine(), x1, x2, len, start_frame, eps, sample)
            return xs, cs, [], b, mu, logstd, pre
        b = self.infer_b(x1, sample, eps=eps)
        xs, cs, zs_gen, _ = self.generate_seq(b[0] if self.ib else b, x2, len, start_frame=start_frame)
        if self.ib:
            b, mu, logstd, pre = b
            return xs, cs, zs_gen, b, mu, logstd, pre
        return xs, cs, zs_gen, b

    def infer_b(self, s, sample, eps=None):
        """:587-601, :175-209.  With the bottleneck: (b, mu, logstd, pre), b = eps exp(logstd) + mu with eps ~ N(0, 1) drawn
        by ``torch.randn`` (``sample``: b is the noise itself, :198-199, :208-209); without: the last hidden state.
        ``eps`` (not in the reference's signature): the noise to use instead of drawing it -- what the parity tests inject."""
        if not self.ib:
            return self.engine().infer_b(s.contiguous(), None)
        if eps is None:
            eps = torch.randn(s.shape[0], self.dim_hidden_b, device=s.device, dtype=s.dtype)
        b, mu, logstd, pre = self.engine().infer_b(s.contiguous(), None if sample else eps)
        return (eps if sample else b), mu, logstd, pre

    def generate_seq(self, b, x_pose, len, start_frame):
        """:603-626: -> (xs [B, len, n_kps], cs (each step's input pose), [], b)."""
        xs, cs = self.engine().generate_seq(b.contiguous(), x_pose.contiguous(), int(len), int(start_frame))
        return xs, cs, [], b
