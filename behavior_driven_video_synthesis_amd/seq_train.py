"""Training plans of the behaviour path (BASELINE config 4; csrc/seq_train.hip, include/vunet_seq_train.h).

The flow stage of ``experiments/behavior_net.py`` (:703-714) is, per step,

    gauss, logdet = latent_flow(bs.detach());  f_loss, log = flow_loss(gauss, logdet)
    flow_optimizer.zero_grad();  f_loss.backward();  flow_optimizer.step()

with ``Adam(lr = flow_lr * batch_size, betas = (0.5, 0.9), weight_decay)`` (:384-395) and ``FlowLoss`` (lib/losses.py:294-317).
``FlowTrainEngine`` runs that step on the kernels of csrc/seq.hip (forward, every layer's output kept) and csrc/seq_train.hip
(loss, input gradients, and the weight gradient fused into Adam's update: dW is never written), recorded once per batch size
into a hipGraph.  The same backward kernels serve autograd (``flow_autograd``): there the gradients are written out and the
reference's own ``torch.optim.Adam`` applies them -- what an unchanged ``main.py`` does through ``dropin``.

There is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional

import torch

from . import _lib
from .ops import _call, _p, _stream
from .seq import ACT_LRELU, FlowEngine, SeqCouplingDesc, _need_device, _up, padded_vector, weight_image  # noqa: F401

LRELU_SLOPE = 0.01   # nn.LeakyReLU() default (lib/modules.py:244)


class SeqAdamHp(ctypes.Structure):
    _fields_ = [("lr_dev", ctypes.c_void_p), ("step_dev", ctypes.c_void_p), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("eps", ctypes.c_float), ("weight_decay", ctypes.c_float), ("resolved_dev", ctypes.c_void_p)]


class SeqDxDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "M", "K", "nets", "S", "ldw")]


class SeqCouplingBwdDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "C", "c1", "ld_g", "ld_in", "ld_out", "ld_full", "Mp", "S", "n_sl", "ld_sl", "c1s")]


class SeqDwLayer(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("w", "m", "v", "g", "bias", "bm", "bv", "bg", "dz", "x")]
                + [(n, ctypes.c_int32) for n in ("M", "K", "ldz", "ldx", "tile0", "tiles_k", "kv", "nchunk", "chunk_z", "chunk_x")]
                + [("dx_raw", ctypes.c_void_p)])


class SeqActnormLayer(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("scale", "loc", "sm", "sv", "lm", "lv", "gs", "gl", "gfull", "out")]
                + [(n, ctypes.c_int32) for n in ("ld", "pad")])


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _table(entries, device) -> torch.Tensor:
    """A ctypes array of descriptors -> a device byte tensor (the kernels read their descriptors from device memory)."""
    arr = (type(entries[0]) * len(entries))(*entries)
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class AdamState:
    """torch.optim.Adam's hyper-parameters with the learning rate (float64) and the step count (int64) on the device."""

    def __init__(self, device, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.lr_dev = torch.full((1,), self.lr, dtype=torch.float64, device=device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.resolved_dev = torch.zeros(2, dtype=torch.float32, device=device)    # written by tick(), read by the step's kernels
        self.hp = SeqAdamHp(self.lr_dev.data_ptr(), self.step_dev.data_ptr(), self.betas[0], self.betas[1], self.eps, self.weight_decay,
                            self.resolved_dev.data_ptr())

    def set_lr(self, lr: float):
        if float(lr) != self.lr:
            self.lr = float(lr)
            self.lr_dev.fill_(self.lr)

    def tick(self):
        _call("vunet_seq_adam_tick", _p(self.step_dev), _p(self.lr_dev), self.hp.beta1, self.hp.beta2, _p(self.resolved_dev), _stream())

    @property
    def step(self) -> int:
        return int(self.step_dev.item())


class _TrainLayer:
    """One Linear layer of one MLP: the (possibly padded) weight / bias images the kernels read, and Adam's moments."""

    def __init__(self, name: str, lin, w_img: torch.Tensor, b_img: torch.Tensor):
        self.name, self.lin, self.w, self.b = name, lin, w_img, b_img
        self.w_inplace = w_img.data_ptr() == lin.weight.data_ptr()
        self.b_inplace = b_img.data_ptr() == lin.bias.data_ptr()
        self.m = self.v = self.bm = self.bv = None

    def moments(self):
        if self.m is None:
            self.m, self.v = torch.zeros_like(self.w), torch.zeros_like(self.w)
            self.bm, self.bv = torch.zeros_like(self.b), torch.zeros_like(self.b)

    def write_back(self):
        """A padded image is the training master: copy its valid part into the module's parameter."""
        if not self.w_inplace:
            m, k = self.lin.weight.shape
            _call("vunet_seq_unpack_rows", _p(self.w), self.w.shape[1], 0, 0, 1, _p(self.lin.weight.data), m, k, 0, _stream())
        if not self.b_inplace:
            n = self.lin.bias.numel()
            _call("vunet_seq_unpack_rows", _p(self.b), self.b.numel(), 0, 0, 1, _p(self.lin.bias.data), 1, n, 0, _stream())


class FlowTrainEngine(FlowEngine):
    """``UnconditionalFlow2`` forward with every intermediate kept, ``FlowLoss``, the backward pass and Adam, for one batch of
    <= 64 rows (``config/behavior_net.yaml``: batch_size 64).  Widths are padded to 64 (the tiles of the backward kernels); a
    parameter whose shape already fits -- every layer of the reference configuration -- is read and updated in place."""

    PAD = 64
    PAD_OUT = 64
    TILED = False     # the parameters are the kernels' images here and change every step: no second copy to keep in step

    def __init__(self, flow, lr: float = 1e-3, betas=(0.5, 0.9), eps: float = 1e-8, weight_decay: float = 0.0):
        super().__init__(flow)
        self._adam_args = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.adam: Optional[AdamState] = None
        self._moments: Dict[str, tuple] = {}
        self._generation = 0
        # the update sweep beside the next half's chain on a second stream: measured slower (8.1 vs 7.75 ms per step: the sweep's
        # 5120 workgroups hold every CU, the chain's 1024-thread workgroups wait for room) -- off unless asked for
        self.two_streams = os.environ.get("VUNET_SEQ_TRAIN_STREAMS", "1") == "2"
        self._side_stream = None
        # hidden activations also handed from layer to layer tile-major (include/vunet_seq_tiled.h; the row-major copies are the
        # backward pass's): bit-identical values
        self.tile_activations = os.environ.get("VUNET_SEQ_TRAIN_TILED_X", "1") != "0"
        # "1": ONE pass over W per layer in the backward pass -- the update sweep also forms the layer's input gradient
        # (vunet_seq_dwx: no second read of W, 2.5 GB less traffic per step).  Built, bit-for-bit tested, and SLOWER: 7.59 ms per
        # step against 7.09 for the separate vunet_seq_dx chain + one vunet_seq_dw launch per coupling half -- the update then runs
        # as 120 dependent per-layer launches of 512 - 1024 workgroups (each ~15 us per 64 x 64 sub-tile, whatever its size) instead
        # of 30 launches of 5120 independent tiles.  Off by default.
        self.fused_dx = os.environ.get("VUNET_SEQ_TRAIN_FUSED_DX", "0") == "1"
        # "1": the slab sum + LeakyReLU' of a hidden layer's input gradient inside the launch that writes the slabs
        # (vunet_seq_dx_finish: the last workgroup of a column stripe to arrive does it; 90 dependent launches fewer per step).
        # Built, bit-for-bit tested, and SLOWER: 9.6 ms per step against 6.5 -- the device-scope release / acquire around the
        # arrival counter (the slabs come from workgroups on other XCDs, i.e. other L2s) writes back and invalidates each L2
        # once per workgroup, in a step whose update sweep keeps every L2 full of dirty lines.  Off by default.
        self.fused_finish = os.environ.get("VUNET_SEQ_TRAIN_FUSED_FINISH", "0") == "1"

    # ---- weights
    def _pack(self, dev=None):
        before = self._packed_for
        super()._pack(dev)
        if self._packed_for == before and getattr(self, "layers", None) is not None:
            return
        dev = self.blocks[0]["scale"].device
        if self.adam is None:
            self.adam = AdamState(dev, **self._adam_args)
        self.layers: List[List[List[List[_TrainLayer]]]] = []    # [block][half][net][layer]
        for bi, (blk, mod) in enumerate(zip(self.blocks, self.flow.sub_layers)):
            per_half = []
            for hi, half in enumerate(blk["halves"]):
                nets = []
                for ni, kind in enumerate(("s", "t")):
                    lins = getattr(mod.coupling, kind)[hi].linears()
                    nets.append([_TrainLayer(f"sub_layers.{bi}.coupling.{kind}.{hi}.main.{2 * li}", lin, half.w[li][ni], half.b[li][ni])
                                 for li, lin in enumerate(lins)])
                per_half.append(nets)
            self.layers.append(per_half)
        self.inv_swap = torch.argsort(self.swap.long()).to(torch.int32)
        # Adam's moments survive a re-pack (a load_state_dict between two steps) as long as the shapes do
        for lay in self._all_layers():
            old = self._moments.get(lay.name)
            if old is not None and old[0].shape == lay.w.shape:
                lay.m, lay.v, lay.bm, lay.bv = old
        self.norm_moments = getattr(self, "norm_moments", None)
        if self.norm_moments is None or self.norm_moments[0].device != dev:
            n = len(self.blocks)
            self.norm_moments = [torch.zeros(n, 4, self.C, device=dev)]   # per block: scale m, scale v, loc m, loc v

    def _all_layers(self):
        return [lay for blk in self.layers for half in blk for net in half for lay in net]

    def _ensure_moments(self):
        for lay in self._all_layers():
            lay.moments()
            self._moments[lay.name] = (lay.m, lay.v, lay.bm, lay.bv)

    # ---- buffers
    def _plan(self, rows: int) -> dict:
        p = self._plans.get(("train", rows))
        if p is not None:
            return p
        dev = self.blocks[0]["scale"].device
        b_pad = _up(rows, 16)
        n = len(self.blocks)
        z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
        halves = [h for blk in self.blocks for h in blk["halves"]]
        h0 = halves[0]
        n_lay = len(h0.dims)
        p = dict(b_pad=b_pad, x_in=z(rows, self.C), x_out=z(rows, self.C), logdet=z(b_pad), noise=z(rows, self.C), scalars=z(4),
                 s0=[z(b_pad, self.ld) for _ in range(n)], s1=[z(b_pad, self.ld) for _ in range(n)],
                 # per block, half: one output buffer per MLP layer (the heads: raw slabs when the head layer splits K)
                 y=[[[z(h.nets * b_pad * m * (h.head_split if li == n_lay - 1 else 1)) for li, (m, _) in enumerate(h.dims)]
                     for h in blk["halves"]] for blk in self.blocks],
                 dz=[[[z(h.nets * b_pad * m) for (m, _) in h.dims] for h in blk["halves"]] for blk in self.blocks],
                 yt=[z(max(h0.nets * b_pad * m for (m, _) in h0.dims[:-1])) for _ in range(2)] if n_lay > 1 else None,
                 dzl=z(b_pad, self.ld), dld=z(b_pad), g0=[z(b_pad, self.ld) for _ in range(2)], g1=z(b_pad, self.ld),
                 gfull=[z(b_pad, self.ld) for _ in range(n)], dx=z(rows, self.C))
        # raw input-gradient slabs: S row ranges of W per launch, chosen so that a launch has >= 256 workgroups
        # fused form (vunet_seq_dwx): one slab per 256-row tile of W; separate form: S row ranges chosen for >= 256 workgroups
        p["tsub"] = [self._dwx_rows(m, k, h0.nets) for (m, k) in h0.dims]      # 64-row sub-tiles per tile of the fused form
        p["S"] = ([(m + 64 * t - 1) // (64 * t) for (m, _), t in zip(h0.dims, p["tsub"])] if self.fused_dx
                  else [self._dx_split(m, k, h0.nets) for (m, k) in h0.dims])
        p["raw"] = z(max(h0.nets * s * b_pad * k for s, (_, k) in zip(p["S"], h0.dims)))
        p["dx_cnt"] = torch.zeros(h0.nets * max(k for (_, k) in h0.dims) // 64, dtype=torch.int32, device=dev)   # vunet_seq_dx_finish
        p["raw_in"] = [z(h0.nets * p["S"][0] * b_pad * h0.dims[0][1]) for _ in range(2)]
        self._plans[("train", rows)] = p
        return p

    @staticmethod
    def _dwx_rows(m: int, k: int, nets: int) -> int:
        """Rows per tile of ``vunet_seq_dwx`` in units of 64: as tall as leaves the launch >= ``VUNET_SEQ_DWX_WGS`` workgroups
        (taller tiles: fewer slabs of dX to write and add; shorter: more workgroups in flight)."""
        want = int(os.environ.get("VUNET_SEQ_DWX_WGS", "512"))     # (measured: 512 -> 7.59 ms per step, 1024 -> 8.16, 2048 -> 8.37)
        t = 4
        while t > 1 and ((m + 64 * t - 1) // (64 * t)) * (k // 64) * nets < want:
            t //= 2
        return t

    @staticmethod
    def _dx_split(m: int, k: int, nets: int) -> int:
        """Row ranges (slabs) of W per input-gradient launch: as many as keep the launch within ONE round of workgroups on the
        chip's 256 CUs and a slab at >= 64 rows, at most 16 -- 4 for the flow's 2048 x 2048 layers (64 stripes), 16 for its first
        layers, 15 for the LSTM's gate matrix over [x | h] (17 stripes: 255 workgroups; slabs of 17 or 18 row groups)."""
        wgs = (k // 64) * nets
        if os.environ.get("VUNET_SEQ_DX_TAIL", "0") == "1":    # the rule before: doubled until >= 256 workgroups (17 stripes: 272)
            s = 1
            while wgs * s < 256 and m % (32 * s) == 0 and m // (2 * s) >= 64 and s < 16:
                s *= 2
            return s
        return max(1, min(16, 256 // wgs, m // 64))

    # ---- forward, everything kept
    def _issue_train_forward(self, rows: int, p: dict):
        """As ``FlowEngine._issue_forward`` (models/flow/blocks.py:540-551, :296-309), into buffers that stay alive."""
        n = len(self.blocks)
        ld_acc = p["logdet"]
        ld_acc.zero_()
        self._step(rows, p["x_in"], self.C, p["s0"][0], self.ld, 0, scale=self.blocks[0]["scale"], loc=self.blocks[0]["loc"], logdet=ld_acc)
        for i, blk in enumerate(self.blocks):
            h0, h1 = blk["halves"]
            tb = p["yt"] if self.tile_activations else None
            part = h0.run(rows, p["s0"][i], self.ld, p["y"][i][0], tile_bufs=tb)
            self._step(rows, p["s0"][i], self.ld, p["s1"][i], self.ld, 0, half=h0, st=part, map_=self.swap, logdet=ld_acc)
            part = h1.run(rows, p["s1"][i], self.ld, p["y"][i][1], tile_bufs=tb)
            last = i == n - 1
            nxt = None if last else self.blocks[i + 1]
            self._step(rows, p["s1"][i], self.ld, p["x_out"] if last else p["s0"][i + 1], self.C if last else self.ld, 0, half=h1, st=part,
                       map_=blk["fwd"], scale=None if last else nxt["scale"], loc=None if last else nxt["loc"], on_src=0, logdet=ld_acc)

    # ---- backward
    def _coupling_bwd(self, rows, p, gbase, ld_g, gslabs, inv_map, scale, in_state, half, heads, gfull, gout, ld_out, dzh):
        h0 = self.blocks[0]["halves"][0]
        d = SeqCouplingBwdDesc(rows, self.C, self.c1, ld_g, self.ld, ld_out, self.ld, half.out_pad if half else self.C,
                               half.head_split if half else 1, (h0.nets * p["S"][0]) if gslabs is not None else 0,
                               h0.dims[0][1], self.c1)
        _call("vunet_seq_coupling_bwd", ctypes.byref(d), _p(gbase), _p(gslabs), _p(inv_map), _p(scale), _p(in_state), _p(heads),
              _p(half.b[-1][0] if half else None), _p(half.b[-1][1] if half else None), _p(p["dld"]), _p(gfull), _p(gout), _p(dzh),
              _stream())

    def _mlp_bwd(self, rows, p, half, ys, dzs, raw_in):
        """The input-gradient chain of one coupling half: head -> first layer.  ``dzs[-1]`` holds the heads' dZ."""
        b_pad = p["b_pad"]
        for li in range(len(half.dims) - 1, -1, -1):
            m_pad, k_pad = half.dims[li]
            s = p["S"][li]
            raw = raw_in if li == 0 else p["raw"]
            d = SeqDxDesc(rows, m_pad, k_pad, half.nets, s, 0)
            w1 = _p(half.w[li][1] if half.nets > 1 else None)
            if li > 0 and self.fused_finish:   # the stripe's last workgroup finishes dZ of the layer below: one launch, not two
                _call("vunet_seq_dx_finish", ctypes.byref(d), _p(half.w[li][0]), w1, _p(dzs[li]), _p(raw), _p(ys[li - 1]),
                      _p(dzs[li - 1]), _p(p["dx_cnt"]), LRELU_SLOPE, _stream())
                continue
            _call("vunet_seq_dx", ctypes.byref(d), _p(half.w[li][0]), w1, _p(dzs[li]), _p(raw), _stream())
            if li > 0:
                _call("vunet_seq_dz_finish", _p(raw), _p(ys[li - 1]), _p(dzs[li - 1]), half.nets, s, b_pad, k_pad, LRELU_SLOPE, _stream())

    def _dw_table(self, p: dict, grads: Optional[dict]) -> dict:
        """The update sweep's descriptor table: per block (last first) and half (1, then 0) the eight layers of its two nets.
        ``grads`` (write mode): name -> (g, bg) buffers."""
        b_pad = p["b_pad"]
        entries, ranges, tile = [], [], 0
        for bi in reversed(range(len(self.blocks))):
            for hi in (1, 0):
                half = self.blocks[bi]["halves"][hi]
                first = tile
                for ni in range(half.nets):
                    for li, (m_pad, k_pad) in enumerate(half.dims):
                        lay = self.layers[bi][hi][ni][li]
                        x = (p["s0"][bi] if hi == 0 else p["s1"][bi]) if li == 0 else p["y"][bi][hi][li - 1][ni * b_pad * k_pad:]
                        dz = p["dz"][bi][hi][li][ni * b_pad * m_pad:]
                        g, bg = grads[lay.name] if grads is not None else (None, None)
                        entries.append(SeqDwLayer(_ptr(lay.w), _ptr(lay.m), _ptr(lay.v), _ptr(g), _ptr(lay.b), _ptr(lay.bm), _ptr(lay.bv),
                                                  _ptr(bg), dz.data_ptr(), x.data_ptr(), m_pad, k_pad, m_pad,
                                                  self.ld if li == 0 else k_pad, tile, k_pad // 64, lay.lin.weight.shape[1], 1, 0, 0, None))
                        tile += (m_pad // 64) * (k_pad // 64)
                ranges.append((first, tile - first))
        return dict(table=_table(entries, self.blocks[0]["scale"].device), n=len(entries), ranges=ranges, tiles=tile)

    def _dwx_table(self, p: dict, grads: Optional[dict]) -> dict:
        """The fused form's table: per block (last first), half (1, then 0) and LAYER (head first) the two nets' entries -- one
        ``vunet_seq_dwx`` launch each: the update of the layer AND its input gradient (raw slabs over the 256-row tiles of W)."""
        b_pad = p["b_pad"]
        entries, ranges, tile = [], {}, 0
        for bi in reversed(range(len(self.blocks))):
            for hi in (1, 0):
                half = self.blocks[bi]["halves"][hi]
                for li in reversed(range(len(half.dims))):
                    m_pad, k_pad = half.dims[li]
                    first = tile
                    s_li = p["S"][li]
                    raw = p["raw_in"][hi] if li == 0 else p["raw"]
                    for ni in range(half.nets):
                        lay = self.layers[bi][hi][ni][li]
                        x = (p["s0"][bi] if hi == 0 else p["s1"][bi]) if li == 0 else p["y"][bi][hi][li - 1][ni * b_pad * k_pad:]
                        dz = p["dz"][bi][hi][li][ni * b_pad * m_pad:]
                        g, bg = grads[lay.name] if grads is not None else (None, None)
                        dx = raw[ni * s_li * b_pad * k_pad:]
                        entries.append(SeqDwLayer(_ptr(lay.w), _ptr(lay.m), _ptr(lay.v), _ptr(g), _ptr(lay.b), _ptr(lay.bm), _ptr(lay.bv),
                                                  _ptr(bg), dz.data_ptr(), x.data_ptr(), m_pad, k_pad, m_pad,
                                                  self.ld if li == 0 else k_pad, tile, k_pad // 64, lay.lin.weight.shape[1], p["tsub"][li], 0, 0,
                                                  dx.data_ptr()))
                        tile += s_li * (k_pad // 64)
                    ranges[(bi, hi, li)] = (first, tile - first)
        return dict(table=_table(entries, self.blocks[0]["scale"].device), n=len(entries), ranges=ranges, tiles=tile)

    def _norm_table(self, p: dict, grads: Optional[dict]) -> torch.Tensor:
        entries = []
        for bi, (blk, mod) in enumerate(zip(self.blocks, self.flow.sub_layers)):
            mom = self.norm_moments[0][bi]
            gs, gl = grads[f"sub_layers.{bi}.norm_layer"] if grads is not None else (None, None)
            entries.append(SeqActnormLayer(_ptr(blk["scale"]), _ptr(blk["loc"]), _ptr(mom[0]), _ptr(mom[1]), _ptr(mom[2]), _ptr(mom[3]),
                                           _ptr(gs), _ptr(gl), _ptr(p["gfull"][bi]), _ptr(p["s0"][bi]), self.ld, 0))
        return _table(entries, self.blocks[0]["scale"].device)

    def _issue_train_backward(self, rows: int, p: dict, gz: torch.Tensor, ld_gz: int, tables: dict, hp: Optional[SeqAdamHp]):
        """``gz``: d loss / d z [rows.., ld_gz]; ``p['dld']``: d loss / d logdet.  Per block, last first: the step that ends the
        block (coupling half 1 + shuffle + the next block's ActNorm), its MLP chain, the update of that half's eight layers;
        the same for half 0; finally block 0's ActNorm and the ActNorm gradients of all blocks."""
        n = len(self.blocks)
        hpp = ctypes.byref(hp) if hp is not None else None
        if self.fused_dx:
            return self._issue_train_backward_fused(rows, p, gz, ld_gz, tables, hpp)
        dw = tables["dw"]
        gbase, ld_g, gsl = gz, ld_gz, None
        rng = iter(dw["ranges"])
        main = torch.cuda.current_stream()
        side = self._side() if self.two_streams else None

        def sweep(first, cnt):
            """The update of one half's eight layers.  Nothing downstream of it in this pass reads what it writes (W, the
            moments) and the chain never touches the dZ / activation buffers it reads again: it runs on the side stream
            beside the next half's input-gradient chain, which alone leaves most of the HBM bandwidth idle."""
            if side is None:
                _call("vunet_seq_dw", _p(dw["table"]), dw["n"], first, cnt, rows, hpp, _stream())
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _call("vunet_seq_dw", _p(dw["table"]), dw["n"], first, cnt, rows, hpp, _stream())
        for i in reversed(range(n)):
            blk = self.blocks[i]
            h0, h1 = blk["halves"]
            last = i == n - 1
            nxt = None if last else self.blocks[i + 1]
            # half 1: out[c] = A_next(v[fwd[c]]), v = couple_1(S1_i)
            self._coupling_bwd(rows, p, gbase, ld_g, gsl, blk["bwd"], None if last else nxt["scale"], p["s1"][i], h1, p["y"][i][1][-1],
                               None if last else p["gfull"][i + 1], p["g1"], self.ld, p["dz"][i][1][-1])
            self._mlp_bwd(rows, p, h1, p["y"][i][1], p["dz"][i][1], p["raw_in"][1])
            sweep(*next(rng))
            # half 0: out[c] = v[swap[c]], v = couple_0(S0_i)
            g0 = p["g0"][i % 2]
            self._coupling_bwd(rows, p, p["g1"], self.ld, p["raw_in"][1], self.inv_swap, None, p["s0"][i], h0, p["y"][i][0][-1], None, g0,
                               self.ld, p["dz"][i][0][-1])
            self._mlp_bwd(rows, p, h0, p["y"][i][0], p["dz"][i][0], p["raw_in"][0])
            sweep(*next(rng))
            gbase, ld_g, gsl = g0, self.ld, p["raw_in"][0]
        # the first step of the pass: S0_0 = A_0(x)
        self._coupling_bwd(rows, p, gbase, ld_g, gsl, None, self.blocks[0]["scale"], None, None, None, p["gfull"][0], p["dx"], self.C, None)
        _call("vunet_seq_actnorm_bwd", _p(tables["norm"]), n, self.C, rows, _p(p["dld"]), hpp, _stream())
        if side is not None:
            main.wait_stream(side)

    def _issue_train_backward_fused(self, rows, p, gz, ld_gz, tables, hpp):
        """The same pass with ONE pass over W per layer: ``vunet_seq_dwx`` updates the layer and leaves its input gradient as raw
        slabs (no ``vunet_seq_dx``: the chain does not read W a second time)."""
        n, b_pad = len(self.blocks), p["b_pad"]
        dwx = tables["dwx"]

        def mlp(bi, hi, half):
            for li in reversed(range(len(half.dims))):
                first, cnt = dwx["ranges"][(bi, hi, li)]
                _call("vunet_seq_dwx", _p(dwx["table"]), dwx["n"], first, cnt, rows, hpp, _stream())
                if li > 0:
                    _call("vunet_seq_dz_finish", _p(p["raw"]), _p(p["y"][bi][hi][li - 1]), _p(p["dz"][bi][hi][li - 1]), half.nets,
                          p["S"][li], b_pad, half.dims[li][1], LRELU_SLOPE, _stream())
        gbase, ld_g, gsl = gz, ld_gz, None
        for i in reversed(range(n)):
            blk = self.blocks[i]
            h0, h1 = blk["halves"]
            last = i == n - 1
            nxt = None if last else self.blocks[i + 1]
            self._coupling_bwd(rows, p, gbase, ld_g, gsl, blk["bwd"], None if last else nxt["scale"], p["s1"][i], h1, p["y"][i][1][-1],
                               None if last else p["gfull"][i + 1], p["g1"], self.ld, p["dz"][i][1][-1])
            mlp(i, 1, h1)
            g0 = p["g0"][i % 2]
            self._coupling_bwd(rows, p, p["g1"], self.ld, p["raw_in"][1], self.inv_swap, None, p["s0"][i], h0, p["y"][i][0][-1], None, g0,
                               self.ld, p["dz"][i][0][-1])
            mlp(i, 0, h0)
            gbase, ld_g, gsl = g0, self.ld, p["raw_in"][0]
        self._coupling_bwd(rows, p, gbase, ld_g, gsl, None, self.blocks[0]["scale"], None, None, None, p["gfull"][0], p["dx"], self.C, None)
        _call("vunet_seq_actnorm_bwd", _p(tables["norm"]), n, self.C, rows, _p(p["dld"]), hpp, _stream())

    def _side(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    def _write_back(self):
        for lay in self._all_layers():
            lay.write_back()

    # ---- the fused step
    def _initialise_actnorm(self, x2: torch.Tensor):
        """ActNorm's data-dependent initialisation happens on the first forward call of a fresh flow (lib/modules.py:303-305)."""
        if self._initialised():
            return
        if not 2 <= x2.shape[0] <= 64:
            raise ValueError("ActNorm's data-dependent initialisation needs one batch of 2..64 rows")
        p = FlowEngine._plan(self, x2.shape[0])
        p["x_in"].copy_(x2)
        self._issue_forward_init(x2.shape[0], p)
        self._all_init = True

    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        _need_device(x)
        _lib.lib()
        self._pack(x.device)
        x2 = x.reshape(x.shape[0], -1)
        if x2.shape[1] != self.C:
            raise ValueError(f"flow over {self.C} channels got {tuple(x.shape)}")
        if not 1 <= x2.shape[0] <= 64:
            raise ValueError(f"the flow's training step takes one batch of 1..64 rows (config/behavior_net.yaml: 64), got {x2.shape[0]}")
        return x2

    def train_step(self, x: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One optimisation step on the batch ``x`` [B, C]: -> device tensor [4] = (flow_loss, reference_nll_loss, nlogdet_loss,
        nll_loss), the entries of ``FlowLoss``'s log (lib/losses.py:310-315).  ``noise``: the draw behind the logged
        ``reference_nll_loss`` (None: ``torch.randn``, as :309 does)."""
        x2 = self._check_input(x.detach())
        rows = x2.shape[0]
        self._initialise_actnorm(x2)
        self._ensure_moments()
        p = self._plan(rows)
        p["x_in"].copy_(x2)
        if noise is None:
            torch.randn(p["noise"].shape, out=p["noise"])
        else:
            p["noise"].copy_(noise.reshape(rows, self.C))
        if "tables" not in p:
            p["tables"] = dict(norm=self._norm_table(p, None), **({"dwx": self._dwx_table(p, None)} if self.fused_dx
                                                                 else {"dw": self._dw_table(p, None)}))

        def issue():
            self.adam.tick()
            self._issue_train_forward(rows, p)
            _call("vunet_seq_flow_loss", _p(p["x_out"]), self.C, _p(p["logdet"]), _p(p["noise"]), rows, self.C, _p(p["scalars"]),
                  _p(p["dzl"]), self.ld, _p(p["dld"]), _stream())
            self._issue_train_backward(rows, p, p["dzl"], self.ld, p["tables"], self.adam.hp)
            self._write_back()
        self.graph.run_step(("train", rows), issue)
        inf = getattr(self.flow, "_engine", None)
        if inf is not None:
            inf._packed_for = None       # its tile-major weight copies are a step behind: re-packed at its next call
        return p["scalars"]

    # ---- autograd's view of the same kernels
    def train_forward(self, x: torch.Tensor):
        x2 = self._check_input(x.detach())
        rows = x2.shape[0]
        self._initialise_actnorm(x2)
        p = self._plan(rows)
        p["x_in"].copy_(x2)
        self._issue_train_forward(rows, p)
        self._generation += 1
        return p["x_out"].clone(), p["logdet"][:rows].clone(), (rows, self._generation)

    def train_backward(self, token, gz: Optional[torch.Tensor], gld: Optional[torch.Tensor]):
        """-> (dx [B, C], {parameter name: gradient}) for the forward pass ``token`` came from."""
        rows, gen = token
        if gen != self._generation:
            raise RuntimeError("the flow was run again before this pass's backward: its saved activations are gone "
                               "(one forward / backward pair at a time)")
        p = self._plan(rows)
        dev = p["x_in"].device
        p["dzl"].zero_()
        if gz is not None:
            p["dzl"][:rows, :self.C].copy_(gz.reshape(rows, self.C))
        p["dld"].zero_()
        if gld is not None:
            p["dld"][:rows].copy_(gld.reshape(rows))
        grads = {lay.name: (torch.empty_like(lay.w), torch.empty_like(lay.b)) for lay in self._all_layers()}
        for bi in range(len(self.blocks)):
            grads[f"sub_layers.{bi}.norm_layer"] = (torch.empty(self.C, device=dev), torch.empty(self.C, device=dev))
        tables = dict(norm=self._norm_table(p, grads), **({"dwx": self._dwx_table(p, grads)} if self.fused_dx
                                                          else {"dw": self._dw_table(p, grads)}))
        self._issue_train_backward(rows, p, p["dzl"], self.ld, tables, None)
        out = {}
        for lay in self._all_layers():
            g, bg = grads[lay.name]
            m, k = lay.lin.weight.shape
            out[lay.name + ".weight"] = g if g.shape == (m, k) else g[:m, :k].contiguous()
            out[lay.name + ".bias"] = bg if bg.numel() == m else bg[:m].contiguous()
        for bi in range(len(self.blocks)):
            gs, gl = grads[f"sub_layers.{bi}.norm_layer"]
            out[f"sub_layers.{bi}.norm_layer.scale"] = gs.reshape(1, self.C, 1, 1)
            out[f"sub_layers.{bi}.norm_layer.loc"] = gl.reshape(1, self.C, 1, 1)
        self._generation += 1      # the activations are spent
        return p["dx"].clone(), out

    # ---- torch.optim.Adam's state layout (checkpoints: experiments/behavior_net.py:392-393, :1003-1011)
    def optimizer_state_dict(self) -> dict:
        self._pack()
        self._ensure_moments()
        step = torch.tensor(float(self.adam.step))
        state, names = {}, [n for n, _ in self.flow.named_parameters()]
        by_name = {}
        for lay in self._all_layers():
            m, k = lay.lin.weight.shape
            by_name[lay.name + ".weight"] = (lay.m[:m, :k], lay.v[:m, :k])
            by_name[lay.name + ".bias"] = (lay.bm[:m], lay.bv[:m])
        for bi in range(len(self.blocks)):
            mom = self.norm_moments[0][bi]
            by_name[f"sub_layers.{bi}.norm_layer.scale"] = (mom[0].reshape(1, -1, 1, 1), mom[1].reshape(1, -1, 1, 1))
            by_name[f"sub_layers.{bi}.norm_layer.loc"] = (mom[2].reshape(1, -1, 1, 1), mom[3].reshape(1, -1, 1, 1))
        for idx, n in enumerate(names):
            m, v = by_name[n]
            state[idx] = {"step": step.clone(), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        a = self.adam
        group = {"lr": a.lr, "betas": a.betas, "eps": a.eps, "weight_decay": a.weight_decay, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "name": "latent_flow",
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict):
        self._pack()
        self._ensure_moments()
        names = [n for n, _ in self.flow.named_parameters()]
        group = sd["param_groups"][0]
        a = self.adam
        a.set_lr(group["lr"])
        a.betas, a.eps, a.weight_decay = tuple(float(b) for b in group["betas"]), float(group["eps"]), float(group["weight_decay"])
        a.hp.beta1, a.hp.beta2, a.hp.eps, a.hp.weight_decay = a.betas[0], a.betas[1], a.eps, a.weight_decay
        lay_by = {lay.name: lay for lay in self._all_layers()}
        step = 0
        for idx, n in enumerate(names):
            st = sd["state"].get(idx)
            if st is None:
                continue
            step = int(st["step"])
            m, v = st["exp_avg"].to(a.lr_dev.device), st["exp_avg_sq"].to(a.lr_dev.device)
            base, leaf = n.rsplit(".", 1)
            if base in lay_by:
                lay = lay_by[base]
                if leaf == "weight":
                    lay.m[:m.shape[0], :m.shape[1]].copy_(m)
                    lay.v[:v.shape[0], :v.shape[1]].copy_(v)
                else:
                    lay.bm[:m.numel()].copy_(m)
                    lay.bv[:v.numel()].copy_(v)
            else:
                bi = int(n.split(".")[1])
                o = 0 if leaf == "scale" else 2
                self.norm_moments[0][bi][o].copy_(m.reshape(-1))
                self.norm_moments[0][bi][o + 1].copy_(v.reshape(-1))
        a.step_dev.fill_(step)
        self.graph.graphs.clear()   # the hyper-parameters are launch arguments of the recorded step


class _FlowFn(torch.autograd.Function):
    """``UnconditionalFlow2.forward(x)`` -> (z, logdet) as one autograd node whose backward is ``FlowTrainEngine.train_backward``:
    what ``f_loss.backward()`` (experiments/behavior_net.py:710) reaches when the reference's own loop drives these modules."""

    @staticmethod
    def forward(ctx, engine, names, x, *params):
        z, logdet, token = engine.train_forward(x)
        ctx.engine, ctx.names, ctx.token, ctx.x_shape = engine, names, token, x.shape
        return z, logdet

    @staticmethod
    def backward(ctx, gz, gld):
        dx, grads = ctx.engine.train_backward(ctx.token, gz, gld)
        out = [None, None, dx.reshape(ctx.x_shape) if ctx.needs_input_grad[2] else None]
        for i, n in enumerate(ctx.names):
            out.append(grads[n] if ctx.needs_input_grad[3 + i] else None)
        return tuple(out)


def flow_autograd(engine: FlowTrainEngine, x: torch.Tensor):
    """(z [B, C], logdet [B]) with a graph: the forward of the flow for a loop that calls ``backward()`` itself."""
    named = list(engine.flow.named_parameters())
    return _FlowFn.apply(engine, [n for n, _ in named], x, *[p for _, p in named])


# ================================================================================================
# the behaviour cVAE (first stage of experiments/behavior_net.py: :591-660)
# ================================================================================================
class SeqCellBwdDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "H", "n", "n_sl", "ld_sl", "hoff_sl", "first", "pad")] + [("gl_stride", ctypes.c_int64)]


from .seq import BehaviorEngine, SeqLinearDesc, SeqLstmDesc, ACT_NONE, linear   # noqa: E402


class BehaviorTrainEngine(BehaviorEngine):
    """``ResidualBehaviorNet.forward(x1, x2, len)`` (models/pose_behavior_rnn.py:574-586) with every step's operand row, cell
    state and gate activations kept, and its backward pass: back-propagation through time over the decoder's roll-out and the
    encoder's LSTM (csrc/seq_bptt.hip + the matrix kernels of csrc/seq_train.hip).  One batch of <= 64 rows
    (config/behavior_net.yaml: 64).  Gradients are written in torch's parameter layout -- into the ``.grad`` views of the fused
    optimiser (``experiments.behavior_net``) or into fresh tensors for autograd (``behavior_autograd``)."""

    HOFF_PAD = 64

    def __init__(self, net):
        super().__init__(net)
        self._generation = 0
        # h handed from LSTM step to LSTM step as tiles too (vunet_seq_lstm_gates_tiled_h; batches of more than 32 rows);
        # "0": the step reads its whole operand from the row-major rows (bit-identical)
        self.tile_h = os.environ.get("VUNET_SEQ_LSTM_TILED_H", "1") != "0"

    def _check(self, dev=None):
        net = self.net
        if net.decoder.use_nin:
            raise NotImplementedError("training with linear_in_decoder=True is not built (config/behavior_net.yaml trains with False): "
                                      "the input layer is folded into the gate matrix on this path")
        if not net.ib:
            raise NotImplementedError("training without the information bottleneck is not built (experiments/behavior_net.py:310 "
                                      "always constructs the net with information_bottleneck=True)")
        self._pack(dev)
        if self.H % 64:
            raise ValueError("dim_hidden_b must be a multiple of 64 for training on the HIP path")

    def _tplan(self, rows: int, t_in: int, t2: int, length: int) -> dict:
        key = ("train", rows, t_in, t2, length)
        p = self._plans.get(key)
        if p is not None:
            return p
        dev = self.dec_w.device
        bp, H, n, ldx = _up(rows, 16), self.H, self.n, self.ldx
        z = lambda *s: torch.zeros(*s, device=dev)   # noqa: E731
        s_d = FlowTrainEngine._dx_split(4 * H, ldx, 1)
        s_e = FlowTrainEngine._dx_split(4 * H, H, 1)
        s_h = FlowTrainEngine._dx_split(H, H, 2)
        p = dict(b_pad=bp, x1=z(rows, t_in, n), x2=z(rows, t2, n), eps=z(rows, H), target=z(rows, length, n),
                 xh_e=z(t_in + 1, bp, ldx), c_e=z(t_in + 1, bp, H), gates_e=z(t_in, bp, 4 * H),
                 xh_d=z(length + 1, bp, ldx), c_d=z(length + 1, bp, H), gates_d=z(length, bp, 4 * H),
                 xraw=z(bp, self.ldraw), pre=z(bp, H), heads=z(2 * bp * H), mu=z(rows, H), logstd=z(rows, H), b=z(rows, H),
                 xs=z(rows, length, n), cs=z(rows, length, n),
                 dgates_d=z(length, bp, 4 * H), dgates_e=z(t_in, bp, 4 * H), gx=z(length + 1, bp, 64), gc=z(bp, H), dy=z(2, bp, H),
                 S=(s_d, s_e, s_h), raw_d=z(s_d * bp * ldx), raw_e=z(s_e * bp * H), raw_h=z(2 * s_h * bp * H),
                 gxs=z(rows, length, n), gmu=z(rows, H), glogstd=z(rows, H),
                 # h handed from step to step as tiles as well (include/vunet_seq_tiled.h: vunet_seq_lstm_gates_tiled_h), > 32 rows
                 ht=[z(bp * H) for _ in range(2)] if (self.tile_h and bp >= 48 and H % 32 == 0 and self.hoff % 32 == 0) else None,
                 g_dec=z(4 * H, ldx), gb_dec=z(4 * H), g_enc=z(4 * H, ldx), gb_enc=z(4 * H), g_out=z(64, H), gb_out=z(64),
                 g_mu=z(H, H), gb_mu=z(H), g_std=z(H, H), gb_std=z(H), zero_bias=z(4 * H),
                 part=z(length + 3 * rows), scalars=z(6), per_seq=z(length))
        # the weight-gradient sweep (write mode): every layer of the net in one launch
        entries, tile = [], 0

        def layer(g, gb, dz, x, m, k, ldz, ldx_, nchunk, cz, cx):
            nonlocal tile
            entries.append(SeqDwLayer(None, None, None, g.data_ptr(), p["zero_bias"].data_ptr(), None, None, gb.data_ptr(), dz.data_ptr(),
                                      x.data_ptr(), m, k, ldz, ldx_, tile, k // 64, k, nchunk, cz, cx, None))
            tile += (m // 64) * (k // 64)
        layer(p["g_dec"], p["gb_dec"], p["dgates_d"], p["xh_d"], 4 * H, ldx, 4 * H, ldx, length, bp * 4 * H, bp * ldx)
        layer(p["g_enc"], p["gb_enc"], p["dgates_e"], p["xh_e"], 4 * H, ldx, 4 * H, ldx, t_in, bp * 4 * H, bp * ldx)
        layer(p["g_out"], p["gb_out"], p["gx"], p["xh_d"][1:, :, self.hoff:], 64, H, 64, ldx, length, bp * 64, bp * ldx)
        layer(p["g_mu"], p["gb_mu"], p["dy"][0], p["pre"], H, H, H, H, 1, 0, 0)
        layer(p["g_std"], p["gb_std"], p["dy"][1], p["pre"], H, H, H, H, 1, 0, 0)
        p["dw"] = dict(table=_table(entries, dev), n=len(entries), tiles=tile)
        self._plans[key] = p
        return p

    # ---- forward, everything kept
    def _issue_train_forward(self, rows, p, t_in, t2, length, start_frame, sample_prior: bool):
        dec = self.net.decoder
        n, H, esz = self.n, self.H, 4
        x1, x2 = p["x1"], p["x2"]
        _call("vunet_seq_start", _p(x1), t_in * n, None, None, None, 0, _p(p["xh_e"][0]), self.ldx, self.hoff, _p(p["c_e"][0]), rows, n, H,
              _stream())
        d = SeqLstmDesc(rows, H, self.ldx, self.hoff, n, self.ldraw, t_in * n)
        ht = p["ht"]
        for t in range(t_in):
            x_next = ctypes.c_void_p(x1.data_ptr() + (t + 1) * n * esz) if t + 1 < t_in else None
            if ht is not None:
                _call("vunet_seq_lstm_gates_tiled_h", ctypes.byref(d), _p(self.enc_wt), _p(p["xh_e"][t]), _p(ht[t & 1]) if t else None,
                      _p(self.enc_b), _p(p["c_e"][t]), _p(p["c_e"][t + 1]), _p(p["xh_e"][t + 1]), _p(ht[(t + 1) & 1]),
                      _p(p["pre"]) if t == t_in - 1 else None, x_next, _p(p["gates_e"][t]), _stream())
                continue
            _call("vunet_seq_lstm_gates_tiled", ctypes.byref(d), _p(self.enc_wt), _p(p["xh_e"][t]), _p(self.enc_b), _p(p["c_e"][t]),
                  _p(p["c_e"][t + 1]), _p(p["xh_e"][t + 1]), _p(p["pre"]) if t == t_in - 1 else None, x_next, _p(p["gates_e"][t]), _stream())
        w, bias = self.heads
        dl = SeqLinearDesc(rows, H, H, H, ACT_NONE, ACT_NONE, 2, 1, 1)
        linear(dl, w, p["pre"], bias, p["heads"])
        _call("vunet_seq_bottleneck", _p(p["heads"]), H, _p(p["eps"]), _p(p["mu"]), _p(p["logstd"]), _p(p["b"]), rows, H, _stream())
        b_used = p["eps"] if sample_prior else p["b"]      # ``sample=True``: b is the noise itself (:198-199, :208-209)
        x0 = ctypes.c_void_p(x2.data_ptr() + start_frame * n * esz)
        _call("vunet_seq_start", x0, t2 * n, _p(b_used), _p(b_used), _p(p["xraw"]), self.ldraw, _p(p["xh_d"][0]), self.ldx, self.hoff,
              _p(p["c_d"][0]), rows, n, H, _stream())
        d = SeqLstmDesc(rows, H, self.ldx, self.hoff, n, self.ldraw, length * n)
        xs, cs = p["xs"], p["cs"]
        for t in range(length):
            if ht is not None:
                _call("vunet_seq_lstm_gates_tiled_h", ctypes.byref(d), _p(self.dec_wt), _p(p["xh_d"][t]), _p(ht[t & 1]) if t else None,
                      _p(self.dec_b), _p(p["c_d"][t]), _p(p["c_d"][t + 1]), _p(p["xh_d"][t + 1]), _p(ht[(t + 1) & 1]), None, None,
                      _p(p["gates_d"][t]), _stream())
            else:
                _call("vunet_seq_lstm_gates_tiled", ctypes.byref(d), _p(self.dec_wt), _p(p["xh_d"][t]), _p(self.dec_b), _p(p["c_d"][t]),
                      _p(p["c_d"][t + 1]), _p(p["xh_d"][t + 1]), None, None, _p(p["gates_d"][t]), _stream())
            _call("vunet_seq_decoder_out", ctypes.byref(d), _p(p["xh_d"][t + 1]), _p(dec.n_out.weight.detach()), _p(dec.n_out.bias.detach()),
                  _p(p["xraw"]), ctypes.c_void_p(xs.data_ptr() + t * n * esz), ctypes.c_void_p(cs.data_ptr() + t * n * esz), _stream())

    # ---- backward
    def _issue_train_backward(self, rows, p, t_in, length, sample_prior: bool, gpre: Optional[torch.Tensor] = None):
        """``p['gxs']`` = d loss / d xs, ``p['gmu']`` / ``p['glogstd']`` = the loss's direct gradients wrt the heads.  Leaves the
        weight gradients as images in ``p`` (``_unpack_grads`` turns them into torch's layout)."""
        dec = self.net.decoder
        n, H, esz = self.n, self.H, 4
        s_d, s_e, s_h = p["S"]
        w_out = dec.n_out.weight.detach()
        for t in reversed(range(length)):
            first = t == length - 1
            d = SeqCellBwdDesc(rows, H, n, s_d, self.ldx, self.hoff, 3 if first else 0, 0, length * n)
            _call("vunet_seq_cell_bwd", ctypes.byref(d), None if first else _p(p["raw_d"]), _p(w_out),
                  ctypes.c_void_p(p["gxs"].data_ptr() + t * n * esz), None if first else _p(p["gx"][t + 1]), _p(p["gx"][t]),
                  _p(p["gates_d"][t]), _p(p["c_d"][t]), _p(p["c_d"][t + 1]), _p(p["gc"]), _p(p["dgates_d"][t]), _stream())
            dd = SeqDxDesc(rows, 4 * H, self.ldx, 1, s_d, 0)
            _call("vunet_seq_dx", ctypes.byref(dd), _p(self.dec_w), None, _p(p["dgates_d"][t]), _p(p["raw_d"]), _stream())
        # h0 = c0 = b (models/pose_behavior_rnn.py:612-614); b = eps exp(logstd) + mu, or the noise itself
        _call("vunet_seq_bottleneck_bwd", _p(p["raw_d"]), s_d, self.ldx, self.hoff, _p(p["gc"]), None if sample_prior else _p(p["eps"]),
              _p(p["logstd"]), _p(p["gmu"]), _p(p["glogstd"]), _p(p["dy"]), rows, H, _stream())
        if sample_prior:   # b does not depend on the encoder: only the loss's direct terms reach the heads
            p["dy"][0, :rows].copy_(p["gmu"])
            p["dy"][1, :rows].copy_(p["glogstd"])
        dh = SeqDxDesc(rows, H, H, 2, s_h, 0)
        _call("vunet_seq_dx", ctypes.byref(dh), _p(self.heads[0][0]), _p(self.heads[0][1]), _p(p["dy"]), _p(p["raw_h"]), _stream())
        if gpre is not None:
            p["raw_h"][:p["b_pad"] * H].view(p["b_pad"], H)[:rows].add_(gpre)
        enc_w_h = ctypes.c_void_p(self.enc_w.data_ptr() + self.hoff * esz)
        for t in reversed(range(t_in)):
            last = t == t_in - 1
            d = SeqCellBwdDesc(rows, H, n, 2 * s_h if last else s_e, H, 0, 2 if last else 0, 0, 0)
            _call("vunet_seq_cell_bwd", ctypes.byref(d), _p(p["raw_h"] if last else p["raw_e"]), None, None, None, None,
                  _p(p["gates_e"][t]), _p(p["c_e"][t]), _p(p["c_e"][t + 1]), _p(p["gc"]), _p(p["dgates_e"][t]), _stream())
            if t > 0:   # (h0 of the encoder is a constant)
                de = SeqDxDesc(rows, 4 * H, H, 1, s_e, self.ldx)
                _call("vunet_seq_dx", ctypes.byref(de), enc_w_h, None, _p(p["dgates_e"][t]), _p(p["raw_e"]), _stream())
        _call("vunet_seq_dw", _p(p["dw"]["table"]), p["dw"]["n"], 0, p["dw"]["tiles"], rows, None, _stream())

    def _unpack_grads(self, p, out: Dict[str, torch.Tensor]):
        """The images of the sweep -> ``out[name]`` (tensors in the parameters' shapes, written in place)."""
        net = self.net
        n, H = self.n, self.H
        _call("vunet_seq_lstm_grads_unpack", _p(p["g_dec"]), _p(p["gb_dec"]), self.ldx, self.hoff, n, H, _p(out["decoder.rnn.weight_ih"]),
              _p(out["decoder.rnn.weight_hh"]), _p(out["decoder.rnn.bias_ih"]), _p(out["decoder.rnn.bias_hh"]), _stream())
        _call("vunet_seq_lstm_grads_unpack", _p(p["g_enc"]), _p(p["gb_enc"]), self.ldx, self.hoff, n, H, _p(out["b_enc.rnn.weight_ih_l0"]),
              _p(out["b_enc.rnn.weight_hh_l0"]), _p(out["b_enc.rnn.bias_ih_l0"]), _p(out["b_enc.rnn.bias_hh_l0"]), _stream())
        _call("vunet_seq_unpack_rows", _p(p["g_out"]), H, 0, 0, 1, _p(out["decoder.n_out.weight"]), n, H, 0, _stream())
        _call("vunet_seq_unpack_rows", _p(p["gb_out"]), 64, 0, 0, 1, _p(out["decoder.n_out.bias"]), 1, n, 0, _stream())
        for tag, head, g, gb in (("mu_fn", net.b_enc.mu_fn, p["g_mu"], p["gb_mu"]), ("std_fn", net.b_enc.std_fn, p["g_std"], p["gb_std"])):
            v, gg, bias, gamma, beta = head._params()
            q = f"b_enc.{tag}."
            _call("vunet_seq_normlinear_bwd", _p(g), _p(gb), _p(v.detach()), _p(gg.detach()), _p(bias.detach()), _p(gamma.detach()), H, H,
                  _p(out[q + "conv.weight_v"]), _p(out[q + "conv.weight_g"]), _p(out[q + "conv.bias"]), _p(out[q + "gamma"]),
                  _p(out[q + "beta"]), _stream())

    def grad_names(self):
        return [n for n, _ in self.net.named_parameters()]

    # ---- autograd's view
    def train_forward(self, x1, x2, length, start_frame, eps, sample_prior):
        _need_device(x1, x2, eps)
        _lib.lib()
        self._check(x1.device)
        rows, t_in, t2 = x1.shape[0], x1.shape[1], x2.shape[1]
        if not 1 <= rows <= 64:
            raise ValueError(f"the behaviour net's training step takes one batch of 1..64 rows, got {rows}")
        if x1.shape[2] != self.n or x2.shape[2] != self.n or x2.shape[0] != rows:
            raise ValueError(f"forward: sequences {tuple(x1.shape)}, {tuple(x2.shape)} for {self.n} pose dimensions")
        start_frame = start_frame % t2
        p = self._tplan(rows, t_in, t2, length)
        p["x1"].copy_(x1)
        p["x2"].copy_(x2)
        p["eps"].copy_(eps)
        self._fill_images()
        self._issue_train_forward(rows, p, t_in, t2, length, start_frame, sample_prior)
        self._generation += 1
        return p, (rows, t_in, t2, length, sample_prior, self._generation)

    def train_backward(self, token, gxs, gcs, gb, gmu, glogstd, gpre):
        rows, t_in, t2, length, sample_prior, gen = token
        if gen != self._generation:
            raise RuntimeError("the net was run again before this pass's backward: its saved activations are gone "
                               "(one forward / backward pair at a time)")
        p = self._tplan(rows, t_in, t2, length)
        p["gxs"].zero_()
        if gxs is not None:
            p["gxs"].copy_(gxs)
        if gcs is not None:      # cs[t] is the input of step t: x_start for t = 0, xs[t - 1] after (:618-620)
            p["gxs"][:, :-1].add_(gcs[:, 1:])
        p["gmu"].zero_()
        p["glogstd"].zero_()
        if gmu is not None:
            p["gmu"].copy_(gmu)
        if glogstd is not None:
            p["glogstd"].copy_(glogstd)
        if gb is not None and not sample_prior:   # b = eps exp(logstd) + mu
            p["gmu"].add_(gb)
            p["glogstd"].add_(gb * p["eps"] * torch.exp(p["logstd"]))
        self._issue_train_backward(rows, p, t_in, length, sample_prior, gpre)
        out = {n: torch.empty_like(q) for n, q in self.net.named_parameters()}
        self._unpack_grads(p, out)
        self._generation += 1
        return out


class _BehaviorFn(torch.autograd.Function):
    """``ResidualBehaviorNet.forward`` as one autograd node (inputs are data: no gradient wrt the sequences)."""

    @staticmethod
    def forward(ctx, engine, names, x1, x2, length, start_frame, eps, sample_prior, *params):
        p, token = engine.train_forward(x1, x2, length, start_frame, eps, sample_prior)
        ctx.engine, ctx.names, ctx.token = engine, names, token
        ctx.set_materialize_grads(False)
        b = p["eps"] if sample_prior else p["b"]
        return p["xs"].clone(), p["cs"].clone(), b.clone(), p["mu"].clone(), p["logstd"].clone(), p["pre"][:x1.shape[0]].clone()

    @staticmethod
    def backward(ctx, gxs, gcs, gb, gmu, glogstd, gpre):
        grads = ctx.engine.train_backward(ctx.token, gxs, gcs, gb, gmu, glogstd, gpre)
        return (None,) * 8 + tuple(grads[n] if ctx.needs_input_grad[8 + i] else None for i, n in enumerate(ctx.names))


def behavior_autograd(engine: BehaviorTrainEngine, x1, x2, length, start_frame, eps, sample_prior):
    """(xs, cs, b, mu, logstd, pre) with a graph whose backward is the HIP back-propagation through time."""
    named = list(engine.net.named_parameters())
    return _BehaviorFn.apply(engine, [n for n, _ in named], x1.contiguous(), x2.contiguous(), int(length), int(start_frame), eps,
                             bool(sample_prior), *[q for _, q in named])
