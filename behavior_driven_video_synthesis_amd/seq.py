"""Launch plans of the behaviour front half of BASELINE config 5 (csrc/seq.hip): flow sample -> pose_behavior_rnn decode.

The reference runs ``flow.reverse`` and ``net.generate_seq`` as ATen chains (experiments/behavior_net.py:1173-1184):
about 600 small launches for the 15-block flow of config/behavior_net.yaml and 250 for a 50-step roll-out.  Here an MLP
is one launch per layer with bias and activation in its epilogue (the s and t nets of a coupling in the same launch),
everything between two MLP evaluations of the flow is one launch, an LSTM step is one launch (the cell update rides in
the gate product's epilogue) plus one for the decoder's output layer, and a whole pass is recorded once per batch size
into a hipGraph and replayed.  ``FlowEngine`` / ``BehaviorEngine`` hold the packed weights and the per-batch-size buffers of one module;
the ``nn.Module`` mirrors (models/flow, models/pose_behavior_rnn.py) own the parameters and call into them.

There is no CPU path: the engines raise without a device tensor and the HIP library.
"""
from __future__ import annotations

import ctypes
import gc
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .ops import _call, _p, _stream

MAX_ROWS = 64   # batch rows per launch (four 16-row MFMA tiles); larger batches are processed in chunks


class SeqLinearDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "M", "K", "ldx", "act0", "act1", "nets", "shared_in", "S")]


class SeqCouplingDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "C", "c1", "ld_in", "ld_out", "Mp", "reverse", "affine_on_src", "S")]


class SeqLstmDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "H", "ldx", "hoff", "n", "ldraw")] + [("seq_stride", ctypes.c_int64)]


def _up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _need_device(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or t.dtype != torch.float32):
            raise RuntimeError("the behaviour path runs on hand-written gfx950 kernels: fp32 device tensors only (no CPU fallback)")


def _need_module_on(dev, named):
    """The kernels take the modules' parameters and buffers as raw pointers: each must be an fp32 (index buffers: integer)
    tensor on the input's device -- a flow left on the CPU, or cast with .double() / .half(), would otherwise be read as
    garbage or fault instead of raising the mismatch torch raises."""
    for name, t in named:
        if t is None:
            continue
        if t.device != dev:
            raise RuntimeError(f"{name} is on {t.device} but the input is on {dev}: move the module with .to(device) first")
        if t.dtype.is_floating_point and t.dtype != torch.float32:
            raise RuntimeError(f"{name} is {t.dtype}: the behaviour path's kernels read fp32 parameters")
        if not t.dtype.is_floating_point and t.dtype not in (torch.int64, torch.int32, torch.uint8):
            raise RuntimeError(f"{name} is {t.dtype}: expected an integer index buffer")


ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2


def weight_image(w: torch.Tensor, m_pad: int, k_pad: int, col_off: int = 0, row_scale: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[M, K] weights -> the zero-padded [m_pad, k_pad] image ``vunet_seq_linear`` reads.  Weights whose shape already
    fits (every layer of the reference configuration) are used in place: no second copy of a 2.5 GB flow."""
    w2 = w.detach().reshape(w.shape[0], -1)
    m, k = w2.shape
    if out is None and row_scale is None and col_off == 0 and (m, k) == (m_pad, k_pad) and w2.is_contiguous():
        return w2
    if out is None:
        out = torch.zeros(m_pad, k_pad, device=w.device, dtype=torch.float32)
    _call("vunet_seq_pack_rows", _p(w2.contiguous()), m, k, _p(row_scale), _p(out), k_pad, col_off, 0, 1, _stream())
    return out


def padded_vector(b: torch.Tensor, n_pad: int) -> torch.Tensor:
    b1 = b.detach().reshape(-1)
    if b1.numel() == n_pad and b1.is_contiguous():
        return b1
    out = torch.zeros(n_pad, device=b.device, dtype=torch.float32)
    _call("vunet_seq_pack_rows", _p(b1.contiguous()), 1, b1.numel(), None, _p(out), n_pad, 0, 0, 1, _stream())
    return out


def linear(desc: SeqLinearDesc, w: Sequence[torch.Tensor], x: torch.Tensor, bias: Sequence[Optional[torch.Tensor]], y: torch.Tensor,
           layout: int = 0, y_rowmajor: Optional[torch.Tensor] = None):
    """``layout`` (include/vunet_seq_tiled.h): 1 = tile-major weight images, 2 = tile-major operand, 4 = tile-major output
    (``y_rowmajor``: and a row-major copy)."""
    if layout:
        _call("vunet_seq_linear_tiled", ctypes.byref(desc), layout, _p(w[0]), _p(w[1] if len(w) > 1 else None), _p(x), _p(bias[0]),
              _p(bias[1] if len(bias) > 1 else None), _p(y), _p(y_rowmajor), _stream())
        return
    _call("vunet_seq_linear", ctypes.byref(desc), _p(w[0]), _p(w[1] if len(w) > 1 else None), _p(x), _p(bias[0]),
          _p(bias[1] if len(bias) > 1 else None), _p(y), _stream())


def tile_image(w_img: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """A row-major [M, K] weight image -> its tile-major copy (``vunet_seq_pack_tiles``): every wave load of the layer's kernel
    then covers 1 KB contiguous."""
    m, k = w_img.shape
    if out is None:
        out = torch.empty(m * k, device=w_img.device, dtype=torch.float32)
    _call("vunet_seq_pack_tiles", _p(w_img), k, m, k, _p(out), _stream())
    return out


class MlpGroup:
    """One or two ``BasicFullyConnectedNet`` stacks (lib/modules.py:236-257) of equal shape evaluated together: one launch per
    layer, bias and activation in the launch's epilogue (LeakyReLU between layers; ``tanh_head[net]``: Tanh after the last).

    ``layers[net]`` = [(weight, bias), ...]; the stacks either share their input (the s and t nets of a coupling) or
    are single."""

    def __init__(self, layers: Sequence[Sequence[Tuple[torch.Tensor, torch.Tensor]]], k_in_pad: int, tanh_head: Sequence[bool],
                 split_head: bool = False, pad_hidden: int = 32, pad_out: int = 16, tiled: bool = False):
        """``pad_hidden`` / ``pad_out``: the multiples the hidden / head widths are padded to (training pads both to 64: the
        tiles of ``vunet_seq_dx`` / ``vunet_seq_dw``).  ``tiled``: the kernels read TILE-MAJOR copies of the weight images and
        hand the hidden activations to each other tile-major (include/vunet_seq_tiled.h) -- a second copy of the weights (the
        row-major parameters stay what ``state_dict`` / an optimiser see), bit-identical results."""
        self.nets = len(layers)
        n_layers = len(layers[0])
        self.dims = []     # per layer: (m_pad, k_pad)
        self.w: List[List[torch.Tensor]] = []
        self.b: List[List[torch.Tensor]] = []
        self.head_act = [ACT_TANH if t else ACT_NONE for t in tanh_head] + [ACT_NONE] * (2 - self.nets)
        k_pad = k_in_pad
        for li in range(n_layers):
            m = layers[0][li][0].shape[0]
            m_pad = _up(m, pad_out) if li == n_layers - 1 else _up(m, pad_hidden)   # a hidden width is the next layer's K
            self.w.append([weight_image(net[li][0], m_pad, k_pad) for net in layers])
            self.b.append([padded_vector(net[li][1], m_pad) for net in layers])
            self.dims.append((m_pad, k_pad))
            k_pad = m_pad
        self.out_pad = self.dims[-1][0]
        self.tiled = tiled and n_layers >= 2
        self.wt = [[tile_image(weight_image(net[li][0], m, k)) for net in layers] for li, (m, k) in enumerate(self.dims)] if self.tiled else None
        # the head layer as raw partial slabs for seq_coupling_kernel to add (``split_head``: the consumer is that kernel): a
        # 512-row head would otherwise run on 32 workgroups per net; K is split until the launch has 256
        self.head_split = 1
        if split_head:
            m_pad, k_pad = self.dims[-1]
            while (self.head_split < 8 and (m_pad // 16) * self.nets * self.head_split < 256
                   and k_pad % (64 * self.head_split) == 0 and k_pad // (2 * self.head_split) >= 512):
                self.head_split *= 2

    def act_floats(self, b_pad: int) -> int:
        return max(self.nets * b_pad * m * (self.head_split if li == len(self.dims) - 1 else 1) for li, (m, _) in enumerate(self.dims))

    def run(self, rows: int, xin: torch.Tensor, ldx: int, bufs: Sequence[torch.Tensor],
            tile_bufs: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """``xin``: [b_pad, ldx] operand (read from column 0).  Returns the buffer holding the heads' [nets][b_pad][out_pad].
        ``bufs``: two buffers used alternately, or one per layer (training keeps every layer's output).  ``tile_bufs`` (training,
        row-major weights): two more buffers through which the hidden activations ALSO go tile-major from layer to layer."""
        src, shared = xin, 1
        last = len(self.dims) - 1
        if tile_bufs is not None and last >= 1:
            for li, (m_pad, k_pad) in enumerate(self.dims):
                dst = bufs[li % len(bufs)]
                act = self.head_act if li == last else [ACT_LRELU, ACT_LRELU]
                d = SeqLinearDesc(rows, m_pad, k_pad, ldx if li == 0 else k_pad, act[0], act[1], self.nets, shared,
                                  self.head_split if li == last else 1)
                if li == last:
                    linear(d, self.w[li], tile_bufs[(li - 1) % 2], self.b[li], dst, layout=2)
                else:
                    linear(d, self.w[li], src if li == 0 else tile_bufs[(li - 1) % 2], self.b[li], tile_bufs[li % 2],
                           layout=4 | (2 if li > 0 else 0), y_rowmajor=dst)
                src, shared = dst, 0
            return src
        for li, (m_pad, k_pad) in enumerate(self.dims):
            dst = bufs[li % len(bufs)]
            act = self.head_act if li == last else [ACT_LRELU, ACT_LRELU]
            d = SeqLinearDesc(rows, m_pad, k_pad, ldx if li == 0 else k_pad, act[0], act[1], self.nets, shared,
                              self.head_split if li == last else 1)
            if self.tiled:   # layer 0 reads the row-major state rows; the heads' output is the coupling kernel's, row-major
                linear(d, self.wt[li], src, self.b[li], dst, layout=1 | (2 if li > 0 else 0) | (4 if li < last else 0))
            else:
                linear(d, self.w[li], src, self.b[li], dst)
            src, shared = dst, 0
        return src


class MlpEngine:
    """A single ``BasicFullyConnectedNet`` on its own: [B, dim] -> [B, out] (rows in chunks of 64)."""

    def __init__(self, net):
        self._net = weakref.ref(net)
        self._packed_for = None

    @property
    def net(self):
        return self._net()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        _need_device(x)
        _lib.lib()
        lin = self.net.linears()
        _need_module_on(x.device, self.net.named_parameters())
        _inference_only([p for l in lin for p in (l.weight, l.bias)])
        key = _versions([p for l in lin for p in (l.weight, l.bias)])
        if key != self._packed_for:
            self.k_pad = _up(lin[0].in_features, 32)
            self.group = MlpGroup([[(l.weight, l.bias) for l in lin]], self.k_pad, [self.net.use_tanh])
            self._packed_for = key
        g, dev, m_out = self.group, x.device, lin[-1].out_features
        outs = []
        for s in range(0, x.shape[0], MAX_ROWS):
            chunk = x[s:s + MAX_ROWS]
            rows, b_pad = chunk.shape[0], _up(chunk.shape[0], 16)
            xin = torch.zeros(b_pad, self.k_pad, device=dev)
            xin[:rows, :chunk.shape[1]].copy_(chunk)
            bufs = [torch.zeros(g.act_floats(b_pad), device=dev) for _ in range(2)]
            y = g.run(rows, xin, self.k_pad, bufs)
            outs.append(y[:b_pad * g.out_pad].view(b_pad, g.out_pad)[:rows, :m_out].contiguous())
        return outs[0] if len(outs) == 1 else torch.cat(outs)


def _flat2(x: torch.Tensor) -> torch.Tensor:
    if x.dim() == 4 and x.shape[2] == 1 and x.shape[3] == 1:
        return x.reshape(x.shape[0], x.shape[1])
    if x.dim() != 2:
        raise ValueError(f"the behaviour path works on [B, C] (or [B, C, 1, 1]) tensors, got {tuple(x.shape)}")
    return x


def actnorm_initialize(mod, x: torch.Tensor):
    """lib/modules.py:270-290 on the rows of ``x`` [B, C]; writes ``mod.loc`` / ``mod.scale`` in place."""
    _need_device(x)
    _need_module_on(x.device, (("ActNorm.loc", mod.loc), ("ActNorm.scale", mod.scale)))
    x2 = _flat2(x).contiguous()
    b, c = x2.shape
    _call("vunet_seq_actnorm_init", _p(x2), c, b, c, _p(mod.loc.data), _p(mod.scale.data), _stream())


def actnorm_apply(mod, x: torch.Tensor, reverse: bool):
    """-> (h shaped like x, logdet [B] or None): lib/modules.py:307-316 / :320-331."""
    _need_device(x)
    _need_module_on(x.device, (("ActNorm.loc", mod.loc), ("ActNorm.scale", mod.scale)))
    _inference_only([mod.loc, mod.scale])
    x2 = _flat2(x).contiguous()
    b, c = x2.shape
    out = torch.empty_like(x2)
    logdet = None if reverse else torch.zeros(b, device=x.device)
    d = SeqCouplingDesc(b, c, c, c, c, c, 1 if reverse else 0, 0, 1)
    _call("vunet_seq_coupling", ctypes.byref(d), _p(x2), None, None, None, None, _p(mod.scale.detach().reshape(-1)),
          _p(mod.loc.detach().reshape(-1)), _p(out), _p(logdet), _stream())
    return out.reshape(x.shape), logdet


class _GraphCache:
    """Record a launch sequence once per key into a hipGraph and replay it (torch.cuda.CUDAGraph is the handle)."""

    MAX_GRAPHS = 16   # per engine; a caller that keeps changing lengths / start frames does not accumulate recordings

    def __init__(self):
        self.graphs: Dict[tuple, torch.cuda.CUDAGraph] = {}
        self.enabled = os.environ.get("VUNET_SEQ_GRAPH", "1") != "0"
        self._stream = None
        self.on_evict = None    # weakref.WeakMethod called with the key of a recording that is dropped (its buffers go too)

    def run(self, key, issue):
        if not self.enabled or torch.cuda.is_current_stream_capturing():   # (inside someone else's capture: become part of it)
            issue()
            return
        g = self.graphs.get(key)
        if g is None:
            issue()                      # once eagerly: first-launch work (module load, function attributes) stays outside
            torch.cuda.synchronize()
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            g = torch.cuda.CUDAGraph()
            # no collector run while the stream records: a finaliser that frees device memory or destroys another graph
            # (an engine of a model that went out of scope) is not a recordable operation and aborts the process
            was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(g, stream=self._stream):
                    issue()
            finally:
                if was_on:
                    gc.enable()
            if len(self.graphs) >= self.MAX_GRAPHS:
                self.drop(next(iter(self.graphs)))      # the oldest recording
            self.graphs[key] = g
        g.replay()


    def run_step(self, key, issue):
        """``run`` for a launch sequence with side effects (an optimisation step): every call executes it exactly once -- the
        first call with a key eagerly (first-launch work stays outside a recording), the second records it and replays the
        recording, later ones replay."""
        if key in self.graphs and self.enabled and not torch.cuda.is_current_stream_capturing():
            self.graphs[key].replay()
            return "replayed"
        warmed = self.__dict__.setdefault("_warmed", set())
        if not self.enabled or torch.cuda.is_current_stream_capturing() or key not in warmed:
            warmed.add(key)
            issue()
            return "eager"
        torch.cuda.synchronize()
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        was_on = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(g, stream=self._stream):
                issue()
        finally:
            if was_on:
                gc.enable()
        if len(self.graphs) >= self.MAX_GRAPHS:
            self.drop(next(iter(self.graphs)))
        self.graphs[key] = g
        g.replay()
        return "captured"

    def drop(self, key):
        if self.graphs.pop(key, None) is not None and self.on_evict is not None:
            cb = self.on_evict()
            if cb is not None:
                cb(key)


def _inference_only(params):
    """The behaviour path has no backward (the training of BASELINE config 4 is not built): a call that autograd would
    record must fail loudly instead of returning tensors without a graph -- the reference's own inference runs under
    ``torch.no_grad()`` (experiments/behavior_net.py:1135)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        raise RuntimeError("the flow / behaviour-net kernels are inference only (no backward is built): call under "
                           "torch.no_grad() or freeze the parameters")


def _versions(params) -> tuple:
    return tuple((p.data_ptr(), p._version) for p in params)


class FlowEngine:
    """``UnconditionalFlow2`` (models/flow/blocks.py:95-128) in both directions for batches of <= 64 rows."""

    PAD = 32          # operand row length / MLP input and hidden widths are padded to this multiple
    PAD_OUT = 16      # ... and the head widths to this one
    TILED = os.environ.get("VUNET_SEQ_TILED", "1") != "0"   # tile-major weight copies + activations (inference plans)

    def __init__(self, flow):
        self._flow = weakref.ref(flow)     # the module owns the engine; no cycle for the collector to find
        self._packed_for = None
        self._plans: Dict[int, dict] = {}
        self.graph = _GraphCache()

    @property
    def flow(self):
        return self._flow()

    # ---- weights
    def _pack(self, dev=None):
        params = list(self.flow.parameters()) + list(self.flow.buffers())
        key = _versions(params)
        if key == self._packed_for:
            return
        if dev is not None:
            _need_module_on(dev, list(self.flow.named_parameters()) + list(self.flow.named_buffers()))
        blocks = list(self.flow.sub_layers)
        c = self.flow.in_channels
        dev = blocks[0].norm_layer.loc.device
        c1 = c // 2 + c % 2
        self.C, self.c1, self.ld = c, c1, _up(c, self.PAD)
        ar = torch.arange(c, device=dev)
        swap = torch.cat(torch.chunk(ar, 2)[::-1])          # x -> cat(chunk(x, 2)[::-1])  (models/flow/blocks.py:301, :314)
        self.swap = swap.to(torch.int32)
        self.blocks = []
        for blk in blocks:
            cp = blk.coupling
            halves = [MlpGroup([[(l.weight, l.bias) for l in cp.s[i].linears()], [(l.weight, l.bias) for l in cp.t[i].linears()]],
                               _up(c1, self.PAD), [True, False], split_head=True, pad_hidden=self.PAD, pad_out=self.PAD_OUT,
                               tiled=self.TILED)
                      for i in range(2)]
            self.blocks.append(dict(
                halves=halves, scale=blk.norm_layer.scale.detach().reshape(-1), loc=blk.norm_layer.loc.detach().reshape(-1),
                fwd=blk.shuffle.forward_shuffle_idx.to(torch.int32), bwd=blk.shuffle.backward_shuffle_idx.to(torch.int32)))
        self._packed_for = key
        self._all_init = False
        # a re-pack that finds every image where it was (parameters used in place, updated by an optimiser: their version
        # counters moved, their storage did not) keeps the buffers and the recordings, which hold nothing but pointers
        sig = tuple(t.data_ptr() for blk in self.blocks for h in blk["halves"] for lay in (h.w + h.b + (h.wt or [])) for t in lay) + tuple(
            blk[k].data_ptr() for blk in self.blocks for k in ("scale", "loc", "fwd", "bwd"))
        if sig != getattr(self, "_image_sig", None):
            self._image_sig = sig
            self._plans.clear()
            self.graph.graphs.clear()
            self.graph.__dict__.pop("_warmed", None)

    def _plan(self, rows: int) -> dict:
        p = self._plans.get(rows)
        if p is None:
            dev = self.blocks[0]["scale"].device
            b_pad = _up(rows, 16)
            halves = [h for blk in self.blocks for h in blk["halves"]]
            nf = max(h.act_floats(b_pad) for h in halves)
            p = dict(b_pad=b_pad,
                     state=[torch.zeros(b_pad, self.ld, device=dev) for _ in range(2)],
                     part=[torch.zeros(nf, device=dev) for _ in range(2)],
                     x_in=torch.zeros(rows, self.C, device=dev), x_out=torch.zeros(rows, self.C, device=dev),
                     logdet=torch.zeros(b_pad, device=dev))
            self._plans[rows] = p
        return p

    # ---- launches
    def _step(self, rows, src, ld_in, dst, ld_out, reverse, half=None, map_=None, scale=None, loc=None, on_src=0, st=None,
              logdet=None):
        d = SeqCouplingDesc(rows, self.C, self.c1, ld_in, ld_out, half.out_pad if half else self.C, reverse, on_src,
                            half.head_split if half else 1)
        _call("vunet_seq_coupling", ctypes.byref(d), _p(src), _p(st), _p(half.b[-1][0] if half else None),
              _p(half.b[-1][1] if half else None), _p(map_), _p(scale), _p(loc), _p(dst), _p(logdet), _stream())

    def _issue_reverse(self, rows: int, p: dict):
        """shuffle^-1, half 1, swap + half 0, ActNorm^-1 per block, last block first (models/flow/blocks.py:552-557, :310-319)."""
        st, cur = p["state"], 0
        n = len(self.blocks)
        self._step(rows, p["x_in"], self.C, st[0], self.ld, 1, map_=self.blocks[n - 1]["bwd"])
        for i in reversed(range(n)):
            blk = self.blocks[i]
            h1, h0 = blk["halves"][1], blk["halves"][0]
            part = h1.run(rows, st[cur], self.ld, p["part"])
            self._step(rows, st[cur], self.ld, st[1 - cur], self.ld, 1, half=h1, st=part, map_=self.swap)
            cur = 1 - cur
            part = h0.run(rows, st[cur], self.ld, p["part"])
            last = i == 0
            self._step(rows, st[cur], self.ld, p["x_out"] if last else st[1 - cur], self.C if last else self.ld, 1, half=h0, st=part,
                       map_=None if last else self.blocks[i - 1]["bwd"], scale=blk["scale"], loc=blk["loc"], on_src=1)
            cur = 1 - cur

    def _issue_forward(self, rows: int, p: dict):
        """ActNorm, half 0, swap + half 1, shuffle per block (models/flow/blocks.py:540-551, :296-309)."""
        st, cur = p["state"], 0
        n = len(self.blocks)
        ld_acc = p["logdet"]
        ld_acc.zero_()
        self._step(rows, p["x_in"], self.C, st[0], self.ld, 0, scale=self.blocks[0]["scale"], loc=self.blocks[0]["loc"], logdet=ld_acc)
        for i in range(n):
            blk = self.blocks[i]
            h0, h1 = blk["halves"]
            part = h0.run(rows, st[cur], self.ld, p["part"])
            self._step(rows, st[cur], self.ld, st[1 - cur], self.ld, 0, half=h0, st=part, map_=self.swap, logdet=ld_acc)
            cur = 1 - cur
            part = h1.run(rows, st[cur], self.ld, p["part"])
            last = i == n - 1
            nxt = None if last else self.blocks[i + 1]
            self._step(rows, st[cur], self.ld, p["x_out"] if last else st[1 - cur], self.C if last else self.ld, 0, half=h1, st=part,
                       map_=blk["fwd"], scale=None if last else nxt["scale"], loc=None if last else nxt["loc"], on_src=0, logdet=ld_acc)
            cur = 1 - cur

    def _issue_forward_init(self, rows: int, p: dict):
        """The forward pass of a flow whose ActNorm layers still await their data-dependent initialisation
        (lib/modules.py:303-305): each block's statistics come from the batch as it reaches that block, so the affine
        step cannot ride on the previous block's last launch.  Issued eagerly, once."""
        st, cur = p["state"], 0
        ld_acc = p["logdet"]
        ld_acc.zero_()
        self._step(rows, p["x_in"], self.C, st[0], self.ld, 0)
        n = len(self.blocks)
        for i, (blk, mod) in enumerate(zip(self.blocks, self.flow.sub_layers)):
            norm = mod.norm_layer
            if norm.initialized.item() == 0:
                _call("vunet_seq_actnorm_init", _p(st[cur]), self.ld, rows, self.C, _p(norm.loc.data), _p(norm.scale.data), _stream())
                norm.initialized.fill_(1)
            self._step(rows, st[cur], self.ld, st[cur], self.ld, 0, scale=blk["scale"], loc=blk["loc"], logdet=ld_acc)
            h0, h1 = blk["halves"]
            part = h0.run(rows, st[cur], self.ld, p["part"])
            self._step(rows, st[cur], self.ld, st[1 - cur], self.ld, 0, half=h0, st=part, map_=self.swap, logdet=ld_acc)
            cur = 1 - cur
            part = h1.run(rows, st[cur], self.ld, p["part"])
            last = i == n - 1
            self._step(rows, st[cur], self.ld, p["x_out"] if last else st[1 - cur], self.C if last else self.ld, 0, half=h1, st=part,
                       map_=blk["fwd"], logdet=ld_acc)
            cur = 1 - cur

    def _run(self, x: torch.Tensor, reverse: bool):
        _need_device(x)
        _inference_only(self.flow.parameters())
        _lib.lib()
        self._pack(x.device)
        if not reverse and not self._initialised():
            x2 = x.reshape(x.shape[0], -1)
            if x2.shape[0] > MAX_ROWS or x2.shape[0] < 2:
                raise ValueError("ActNorm's data-dependent initialisation needs one batch of 2..64 rows")
            p = self._plan(x2.shape[0])
            p["x_in"].copy_(x2)
            self._issue_forward_init(x2.shape[0], p)
            return p["x_out"].clone(), p["logdet"][:x2.shape[0]].clone()
        x2 = x.reshape(x.shape[0], -1)
        if x2.shape[1] != self.C:
            raise ValueError(f"flow over {self.C} channels got {tuple(x.shape)}")
        if x2.shape[0] == 0:
            return x2.clone() if reverse else (x2.clone(), x2.new_zeros(0))
        outs, lds = [], []
        for s in range(0, x2.shape[0], MAX_ROWS):
            chunk = x2[s:s + MAX_ROWS]
            rows = chunk.shape[0]
            p = self._plan(rows)
            p["x_in"].copy_(chunk)
            self.graph.run((rows, reverse), (lambda: self._issue_reverse(rows, p)) if reverse else (lambda: self._issue_forward(rows, p)))
            outs.append(p["x_out"].clone())
            if not reverse:
                lds.append(p["logdet"][:rows].clone())
        out = outs[0] if len(outs) == 1 else torch.cat(outs)
        if reverse:
            return out
        return out, (lds[0] if len(lds) == 1 else torch.cat(lds))

    def _initialised(self) -> bool:
        if not getattr(self, "_all_init", False):
            self._all_init = all(int(b.norm_layer.initialized.item()) != 0 for b in self.flow.sub_layers)
        return self._all_init

    def reverse(self, z: torch.Tensor) -> torch.Tensor:
        return self._run(z, True)

    def forward(self, x: torch.Tensor):
        return self._run(x, False)


class BehaviorEngine:
    """``ResidualBehaviorNet`` (models/pose_behavior_rnn.py:538-626): the behaviour encoder's LSTM + bottleneck heads and the
    residual decoder's roll-out, for batches of <= 64 rows."""

    def __init__(self, net):
        self._net = weakref.ref(net)
        self._packed_for = None
        self._plans: Dict[int, dict] = {}
        self.graph = _GraphCache()
        self.graph.on_evict = weakref.WeakMethod(self._drop_io)   # (no cycle: the engine's recordings are freed by refcount)

    def _drop_io(self, key):
        """A recording and the input / output buffers its launches point at live and die together."""
        for p in self._plans.values():
            if "io" in p:          # (the training plans of seq_train.BehaviorTrainEngine keep their buffers for good)
                p["io"].pop(key, None)

    def _io(self, p: dict, key, make):
        io = p["io"].get(key)
        if io is None:
            if len(p["io"]) >= _GraphCache.MAX_GRAPHS:
                old = next(iter(p["io"]))
                self.graph.drop(old)
                p["io"].pop(old, None)
            io = p["io"][key] = make()
        return io

    @property
    def net(self):
        return self._net()

    HOFF_PAD = 32     # the hidden part of an operand row [x | 0 | h] starts at this multiple (training: 64, the backward tiles)

    def _pack(self, dev=None):
        net = self.net
        key = _versions(list(net.parameters()))
        if key == self._packed_for:
            return
        if dev is not None:
            _need_module_on(dev, net.named_parameters())
        dec, enc = net.decoder, net.b_enc
        self.n, self.H = dec.n_in_out, dec.n_hidden
        if self.H % 32:
            raise ValueError("dim_hidden_b must be a multiple of 32 on the HIP path")
        self.hoff = _up(self.n, self.HOFF_PAD)
        self.ldx = self.hoff + self.H
        self.ldraw = _up(self.n, 64)
        dev = dec.rnn.weight_ih.device
        z = lambda *shape: torch.zeros(*shape, device=dev)   # noqa: E731
        # the images: allocated once per geometry, refilled when the parameters change (zero padding is written once)
        geo = (self.n, self.H, self.hoff, dec.use_nin, bool(enc.ib), str(dev))
        if getattr(self, "_geo", None) != geo:
            self._geo = geo
            self.dec_w, self.enc_w = z(4 * self.H, self.ldx), z(4 * self.H, self.ldx)
            # tile-major copies of the two gate images (include/vunet_seq_tiled.h): what the step kernel reads
            self.dec_wt, self.enc_wt = z(4 * self.H * self.ldx), z(4 * self.H * self.ldx)
            self.dec_b, self.enc_b = z(4 * self.H), z(4 * self.H)
            self.dec_fold = (z(4 * self.H, self.n), z(4 * self.H)) if dec.use_nin else None
            self.heads = ([z(self.H, self.H) for _ in range(2)], [z(self.H) for _ in range(2)]) if enc.ib else None
            self.head_rs = [z(self.H) for _ in range(2)] if enc.ib else None
            self._plans.clear()
            self.graph.graphs.clear()
            self.graph.__dict__.pop("_warmed", None)
        self._fill_images()
        self._packed_for = key

    def _fill_images(self):
        """(Re)write the kernels' weight images from the module's parameters, in place."""
        net = self.net
        dec, enc = net.decoder, net.b_enc

        def gate_image(w_ih, w_hh, img):
            """[W_ih | 0 | W_hh] with gate-interleaved rows: row 4 j + q = gate q (i, f, g, o) of hidden unit j."""
            for src, off in ((w_ih.detach().contiguous(), 0), (w_hh.detach().contiguous(), self.hoff)):
                for q in range(4):
                    part = src[q * self.H:(q + 1) * self.H]
                    _call("vunet_seq_pack_rows", _p(part), self.H, part.shape[1], None, _p(img), self.ldx, off, q, 4, _stream())

        def gate_bias(b_ih, b_hh, fold, out):
            _call("vunet_seq_lstm_bias", _p(b_ih.detach().contiguous()), _p(b_hh.detach().contiguous()), _p(fold), self.H, _p(out),
                  _stream())
        self.dec_fold_bias = None
        w_ih = dec.rnn.weight_ih
        if dec.use_nin:   # x = n_in(x) in front of the cell (:494-495): W_ih (W_in x + b_in) = (W_ih W_in) x + W_ih b_in
            w_fold, self.dec_fold_bias = self.dec_fold
            _call("vunet_seq_fold_input", _p(w_ih.detach().contiguous()), _p(dec.n_in.weight.detach().contiguous()),
                  _p(dec.n_in.bias.detach().contiguous()), 4 * self.H, self.n, _p(w_fold), _p(self.dec_fold_bias), _stream())
            w_ih = w_fold
        gate_image(w_ih, dec.rnn.weight_hh, self.dec_w)
        gate_bias(dec.rnn.bias_ih, dec.rnn.bias_hh, self.dec_fold_bias, self.dec_b)
        gate_image(enc.rnn.weight_ih_l0, enc.rnn.weight_hh_l0, self.enc_w)
        gate_bias(enc.rnn.bias_ih_l0, enc.rnn.bias_hh_l0, None, self.enc_b)
        tile_image(self.dec_w, out=self.dec_wt)
        tile_image(self.enc_w, out=self.enc_wt)
        if enc.ib:
            for i, head in enumerate((enc.mu_fn, enc.std_fn)):
                v, g, b, gamma, beta = head._params()
                _call("vunet_seq_normlinear_rows", _p(v.detach().contiguous()), _p(g.detach().contiguous()), _p(b.detach().contiguous()),
                      _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()), self.H, self.H, _p(self.head_rs[i]),
                      _p(self.heads[1][i]), _stream())
                weight_image(v, self.H, self.H, 0, row_scale=self.head_rs[i], out=self.heads[0][i])

    def _plan(self, rows: int) -> dict:
        p = self._plans.get(rows)
        if p is None:
            dev = self.dec_w.device
            b_pad = _up(rows, 16)
            p = dict(b_pad=b_pad, xh=[torch.zeros(b_pad, self.ldx, device=dev) for _ in range(2)],
                     c=[torch.zeros(b_pad, self.H, device=dev) for _ in range(2)],
                     xraw=torch.zeros(b_pad, self.ldraw, device=dev),
                     pre=torch.zeros(b_pad, self.H, device=dev), b_in=torch.zeros(rows, self.H, device=dev),
                     heads=torch.zeros(2 * b_pad * self.H, device=dev), io={})
            # more than 32 rows: h also goes from step to step as tiles (include/vunet_seq_tiled.h: vunet_seq_lstm_gates_tiled_h)
            tile_h = (os.environ.get("VUNET_SEQ_LSTM_TILED_H", "1") != "0" and b_pad >= 48 and self.H % 32 == 0
                      and self.hoff % 32 == 0 and self.ldx == self.hoff + self.H)
            p["ht"] = [torch.zeros(b_pad * self.H, device=dev) for _ in range(2)] if tile_h else None
            self._plans[rows] = p
        return p

    def _lstm_step(self, d, wt, bias, p, t, h_out, x_next):
        """Step t of a roll-out on the plan's ping-pong buffers (t even: 0 -> 1)."""
        cur, nxt = t % 2, 1 - t % 2
        ht = p["ht"]
        if ht is not None:
            _call("vunet_seq_lstm_gates_tiled_h", ctypes.byref(d), _p(wt), _p(p["xh"][cur]), _p(ht[cur]) if t else None, _p(bias),
                  _p(p["c"][cur]), _p(p["c"][nxt]), _p(p["xh"][nxt]), _p(ht[nxt]), h_out, x_next, None, _stream())
        else:
            _call("vunet_seq_lstm_gates_tiled", ctypes.byref(d), _p(wt), _p(p["xh"][cur]), _p(bias), _p(p["c"][cur]), _p(p["c"][nxt]),
                  _p(p["xh"][nxt]), h_out, x_next, None, _stream())

    def _issue_decode(self, rows, p, x_pose, t_in, start_frame, length, xs, cs):
        dec = self.net.decoder
        n, esz = self.n, 4
        x0 = ctypes.c_void_p(x_pose.data_ptr() + start_frame * n * esz)
        _call("vunet_seq_start", x0, t_in * n, _p(p["b_in"]), _p(p["b_in"]), _p(p["xraw"]), self.ldraw, _p(p["xh"][0]), self.ldx,
              self.hoff, _p(p["c"][0]), rows, n, self.H, _stream())
        d = SeqLstmDesc(rows, self.H, self.ldx, self.hoff, n, self.ldraw, length * n)
        for t in range(length):
            nxt = 1 - t % 2
            self._lstm_step(d, self.dec_wt, self.dec_b, p, t, None, None)
            _call("vunet_seq_decoder_out", ctypes.byref(d), _p(p["xh"][nxt]), _p(dec.n_out.weight.detach()), _p(dec.n_out.bias.detach()),
                  _p(p["xraw"]), ctypes.c_void_p(xs.data_ptr() + t * n * esz), ctypes.c_void_p(cs.data_ptr() + t * n * esz), _stream())

    def generate_seq(self, b: torch.Tensor, x_pose: torch.Tensor, length: int, start_frame: int):
        """-> (xs [B, len, n], cs [B, len, n]); models/pose_behavior_rnn.py:603-626."""
        _need_device(b, x_pose)
        _inference_only(self.net.parameters())
        _lib.lib()
        self._pack(b.device)
        if x_pose.dim() != 3 or x_pose.shape[2] != self.n or b.shape[1] != self.H:
            raise ValueError(f"generate_seq: poses {tuple(x_pose.shape)}, behaviour {tuple(b.shape)}")
        start_frame = start_frame % x_pose.shape[1]
        xs_all, cs_all = [], []
        for s in range(0, b.shape[0], MAX_ROWS):
            bc, xc = b[s:s + MAX_ROWS], x_pose[s:s + MAX_ROWS]
            rows, t_in = bc.shape[0], xc.shape[1]
            p = self._plan(rows)
            key = ("dec", rows, t_in, start_frame, length)
            dev = b.device
            io = self._io(p, key, lambda: dict(x=torch.zeros(rows, t_in, self.n, device=dev),
                                               xs=torch.zeros(rows, length, self.n, device=dev),
                                               cs=torch.zeros(rows, length, self.n, device=dev)))
            io["x"].copy_(xc)
            p["b_in"].copy_(bc)
            self.graph.run(key, lambda: self._issue_decode(rows, p, io["x"], t_in, start_frame, length, io["xs"], io["cs"]))
            xs_all.append(io["xs"].clone())
            cs_all.append(io["cs"].clone())
        if len(xs_all) == 1:
            return xs_all[0], cs_all[0]
        return torch.cat(xs_all), torch.cat(cs_all)

    def _issue_encode(self, rows, p, seq, t_in, eps, mu, logstd, b_out):
        n, esz = self.n, 4
        _call("vunet_seq_start", _p(seq), t_in * n, None, None, None, 0, _p(p["xh"][0]), self.ldx, self.hoff, _p(p["c"][0]), rows, n,
              self.H, _stream())
        d = SeqLstmDesc(rows, self.H, self.ldx, self.hoff, n, self.ldraw, t_in * n)
        for t in range(t_in):   # one launch per time step: gate product, cell update and the next input row
            x_next = ctypes.c_void_p(seq.data_ptr() + (t + 1) * n * esz) if t + 1 < t_in else None
            self._lstm_step(d, self.enc_wt, self.enc_b, p, t, _p(p["pre"]) if t == t_in - 1 else None, x_next)
        if self.heads is not None:
            w, bias = self.heads
            dl = SeqLinearDesc(rows, self.H, self.H, self.H, ACT_NONE, ACT_NONE, 2, 1, 1)
            linear(dl, w, p["pre"], bias, p["heads"])
            _call("vunet_seq_bottleneck", _p(p["heads"]), self.H, _p(eps), _p(mu), _p(logstd), _p(b_out), rows, self.H, _stream())

    def infer_b(self, seq: torch.Tensor, eps: Optional[torch.Tensor]):
        """LSTM over ``seq`` [B, T, n] from a zero state.  -> (b, mu, logstd, pre) with the bottleneck heads (b = eps *
        exp(logstd) + mu; eps None: b = mu), else pre   (models/pose_behavior_rnn.py:175-201, :587-601)."""
        _need_device(seq, eps)
        _inference_only(self.net.parameters())
        _lib.lib()
        self._pack(seq.device)
        if seq.dim() != 3 or seq.shape[2] != self.n:
            raise ValueError(f"infer_b: sequence {tuple(seq.shape)}")
        res = []
        for s in range(0, seq.shape[0], MAX_ROWS):
            sc = seq[s:s + MAX_ROWS]
            rows, t_in = sc.shape[0], sc.shape[1]
            p = self._plan(rows)
            key = ("enc", rows, t_in, eps is not None)
            dev = seq.device
            io = self._io(p, key, lambda: dict(x=torch.zeros(rows, t_in, self.n, device=dev), eps=torch.zeros(rows, self.H, device=dev),
                                               mu=torch.zeros(rows, self.H, device=dev), logstd=torch.zeros(rows, self.H, device=dev),
                                               b=torch.zeros(rows, self.H, device=dev)))
            io["x"].copy_(sc)
            if eps is not None:
                io["eps"].copy_(eps[s:s + MAX_ROWS])
            self.graph.run(key, lambda: self._issue_encode(rows, p, io["x"], t_in, io["eps"] if eps is not None else None, io["mu"],
                                                           io["logstd"], io["b"]))
            pre = p["pre"][:rows].clone()
            res.append((io["b"].clone(), io["mu"].clone(), io["logstd"].clone(), pre) if self.heads is not None else (pre,))
        if self.heads is None:
            return res[0][0] if len(res) == 1 else torch.cat([r[0] for r in res])
        if len(res) == 1:
            return res[0]
        return tuple(torch.cat([r[i] for r in res]) for i in range(4))
