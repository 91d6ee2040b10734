"""torch.autograd glue over the C-ABI HIP library (include/vunet_hip.h).

PyTorch is plumbing here: it owns device memory, streams and the autograd tape; every convolution,
normalisation, reduction, loss term and optimiser update of the hot path is a kernel of libvunet_hip.so,
called through ctypes with raw device pointers.  What PyTorch itself still launches inside a training
step (measured, tools/aten_in_step.py: 0.85 ms of a 29 ms step): the autograd engine's accumulation of
fan-out gradients (~29 `add_` per step -- skip tensors with two consumers; fusing them into the data-gradient
epilogue would mean writing into buffers the engine owns), gradient-bucket / scalar zero-fills, the two
`randn_like` draws of the posterior noise and a dozen one-element scalar ops of the loss assembly and the
gamma controller.  There is no CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import contextlib
import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

ACT_NONE, ACT_ELU, ACT_RELU, ACT_SIGMOID, ACT_LRELU = 0, 1, 2, 3, 4
SEED2_OFFSET = 0x9E3779B9  # seed offset of the second source of a dual-source conv (csrc/conv_gather.hip)


class ConvDesc(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("C1", ctypes.c_int32), ("C2", ctypes.c_int32), ("Hs", ctypes.c_int32),
                ("Ws", ctypes.c_int32), ("M", ctypes.c_int32), ("m_off", ctypes.c_int32), ("Mpad", ctypes.c_int32),
                ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("KH", ctypes.c_int32), ("KW", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("mode", ctypes.c_int32),
                ("in_act", ctypes.c_int32), ("in_slope", ctypes.c_float), ("drop_p", ctypes.c_float),
                ("drop_seed", ctypes.c_uint32), ("out_act", ctypes.c_int32), ("d2s", ctypes.c_int32),
                ("aux_act", ctypes.c_int32), ("aux_slope", ctypes.c_float), ("aux_drop_p", ctypes.c_float),
                ("aux_drop_seed", ctypes.c_uint32)]


class WgradDesc(ctypes.Structure):
    _fields_ = [("N", ctypes.c_int32), ("C1", ctypes.c_int32), ("C2", ctypes.c_int32), ("Hs", ctypes.c_int32),
                ("Ws", ctypes.c_int32), ("Cout", ctypes.c_int32), ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32),
                ("KH", ctypes.c_int32), ("KW", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("in_act", ctypes.c_int32), ("in_slope", ctypes.c_float), ("drop_p", ctypes.c_float),
                ("drop_seed", ctypes.c_uint32), ("nsplit", ctypes.c_int32), ("flags", ctypes.c_int32)]


class WgradItem(ctypes.Structure):   # vunet_wgrad_item: one layer of a batched weight-gradient launch
    _fields_ = [("d", WgradDesc)] + [(n, ctypes.c_void_p) for n in ("x1", "x2", "dy", "slabs", "dshift", "amax_x", "amax_x2",
                                                                   "amax_dy")]


class WnDesc(ctypes.Structure):
    _fields_ = [("Cout", ctypes.c_int32), ("C1", ctypes.c_int32), ("C2", ctypes.c_int32), ("KH", ctypes.c_int32),
                ("KW", ctypes.c_int32), ("kind", ctypes.c_int32), ("split", ctypes.c_int32)]


class WnBwdItem(ctypes.Structure):
    _fields_ = ([(n, ctypes.c_void_p) for n in ("slabs", "dshift", "v", "g", "bias", "gamma", "invnorm", "dv", "dg", "dbias",
                                                "dgamma", "dbeta", "workspace")] +
                [("d", WnDesc), ("nsplit", ctypes.c_int32), ("accumulate", ctypes.c_int32)])


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # the raw handle of torch's current stream (torch.cuda.current_stream() builds a Stream object per call: ~10 us,
    # measured 5 ms of host time per training step over ~500 launches)
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vunet HIP path needs device tensors (hand-written gfx950 kernels; no CPU fallback)")
        if t is not None and t.dtype != torch.float32:
            raise RuntimeError(f"vunet HIP path computes in fp32, got {t.dtype}")


def _call(name: str, *args):
    rc = getattr(_lib.lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed with code {rc}")


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else (t if t.is_contiguous() else t.contiguous())


def _r2(x: int) -> int:
    return (x + 1) // 2 * 2


def _r32(x: int) -> int:
    return (x + 31) // 32 * 32


# ------------------------------------------------------------------------------------------------
# dropout seeds: one fresh 32-bit seed per fused conv call, derived from torch's seed
# ------------------------------------------------------------------------------------------------
_drop_state = {"base": None, "ctr": 0, "step": None, "host_step": 0}


def set_dropout_seed(seed: int):
    _drop_state["base"] = int(seed) & 0xFFFFFFFF
    _drop_state["ctr"] = 0


def set_dropout_host_step(step: int):
    """The step's number as a HOST value: every seed handed out afterwards is offset by ``step * 0x9E3779B1`` -- what the
    kernels add themselves from the device counter of ``set_dropout_step``.  A trainer that restarts the seed sequence
    every step (``reset_dropout_counter``) and sets this draws the same masks and the same posterior noise as the
    device-schedule / captured step of the same number (0: no offset)."""
    _drop_state["host_step"] = int(step) & 0xFFFFFFFF


def reset_dropout_counter():
    """Restart the per-call seed sequence: the n-th fused conv of every step gets the same seed again.  Used together
    with ``set_dropout_step``: the step-to-step variation of the masks then comes from the device counter alone, so the
    launch arguments of a step are constant and the step can be replayed from a captured hipGraph."""
    if _drop_state["base"] is None:
        set_dropout_seed(torch.initial_seed())
    _drop_state["ctr"] = 0


def set_dropout_step(counter: Optional[torch.Tensor]):
    """Process-wide device-resident dropout step counter (one int32 element on the GPU, read as uint32; None clears it):
    every dropout prologue launched while it is set hashes with ``seed + counter * 0x9E3779B1`` (vunet_set_dropout_step)."""
    if counter is not None and not (counter.is_cuda and counter.dtype == torch.int32 and counter.numel() == 1):
        raise RuntimeError("dropout step counter: one int32 element on the GPU")
    _drop_state["step"] = counter   # keeps the allocation alive while the library points at it
    _call("vunet_set_dropout_step", _p(counter))


def dropout_step_counter() -> Optional[torch.Tensor]:
    """The device counter the library currently hashes with (None: plain per-call seeds)."""
    return _drop_state["step"]


def next_dropout_seed() -> int:
    if _drop_state["base"] is None:
        set_dropout_seed(torch.initial_seed())
    _drop_state["ctr"] += 1
    x = (_drop_state["base"] * 0x9E3779B1 + _drop_state["ctr"] * 0x85EBCA77) & 0xFFFFFFFF
    x ^= x >> 15
    return ((x * 0x2C1B3C6D) + _drop_state["host_step"] * 0x9E3779B1) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# fused (weight-normalised) convolution
# ------------------------------------------------------------------------------------------------
@dataclass
class ConvCfg:
    kind: int = 0          # 0 NormConv2d, 1 plain conv, 2 L2NormConv2d
    k: int = 3
    stride: int = 1
    pad: int = 0
    in_act: int = ACT_NONE
    in_slope: float = 0.0
    drop_p: float = 0.0
    drop_seed: int = 0
    out_act: int = ACT_NONE
    d2s: bool = False
    res_is_x1: bool = False  # residual tensor is source 1 itself -> its gradient is fused into the dgrad epilogue
    owner: object = None     # the calling module (records its source split; looked up in the active prepack set)
    bf16: bool = False       # set by fused_conv: caller is under no_grad inside ops.inference_precision("bf16")
    passthrough: bool = False   # also return x1 (an alias): its gradient is added in the data-gradient epilogue


def conv_out_size(h: int, k: int, stride: int, pad: int) -> int:
    return (h + 2 * pad - k) // stride + 1


# ------------------------------------------------------------------------------------------------
# convolution arithmetic, process-wide.  All three are fp32-accurate (tests/test_hip_x6.py, tools/x6_accuracy.py):
#   "h2"  operands scaled by a power of two and split into two fp16 terms, three partial products on the fp16 matrix
#         cores wherever the split kernels apply (csrc/conv_h2_kernel.h) -- the default;
#   "x6"  operands split into three bf16 terms, six partial products on the bf16 matrix cores (csrc/conv_x6_kernel.h):
#         no data-dependent scale, twice the matrix instructions;
#   "f32" fp32-input MFMA kernels everywhere.
# The environment variable VUNET_CONV_PRECISION=f32 pins the library itself to f32 regardless of this switch.
# ------------------------------------------------------------------------------------------------
_SCHEMES = {"f32": 0, "x6": 1, "h2": 2}
_conv_precision = {"mode": os.environ.get("VUNET_CONV_SCHEME", "h2")}


def set_conv_precision(mode: str):
    if mode not in _SCHEMES:
        raise ValueError(f"unknown conv precision {mode!r} (h2 | x6 | f32)")
    _conv_precision["mode"] = mode


def conv_precision() -> str:
    return _conv_precision["mode"]


def _scheme() -> int:
    return _SCHEMES[_conv_precision["mode"]]


_wants_split_cache = {}
_TUNING_KEYS = {"split_force_nt": 0, "tiled_force_nt": 1, "force_small": 2, "blk_force_nt": 3, "parity_launches": 4, "blk_ws": 5, "wgrad_rowsplit": 6, "s2_fwd_f32": 7, "p2_form": 8}


def set_tuning(key: str, value: int):
    """vunet_set_tuning: process-wide dispatcher overrides for tests / kernel tuning (0 = the dispatcher's own choice)."""
    _call("vunet_set_tuning", _TUNING_KEYS[key], int(value))
    _wants_split_cache.clear()


def apply_env_tuning():
    """``VUNET_TUNING="key=value,key=value"`` (keys of ``_TUNING_KEYS``): dispatcher overrides for A/B runs of whole programs
    (bench.py calls this once; tests use ``set_tuning``)."""
    spec = os.environ.get("VUNET_TUNING", "")
    for kv in filter(None, (t.strip() for t in spec.split(","))):
        k, _, v = kv.partition("=")
        set_tuning(k.strip(), int(v))


def absmax_partials(x1, x2=None):
    """vunet_absmax_partials: the 1024 partial |x| maxima (512 per source) the split-fp16 kernels scale their input by."""
    out = torch.empty(1024, device=x1.device, dtype=torch.float32)
    _call("vunet_absmax_partials", _p(x1), x1.numel(), _p(x2), 0 if x2 is None else x2.numel(), _p(out), _stream())
    return out


# ---- |y| maxima published by the producer.  The fp16 split kernels can leave the partial maxima of what they store in a
# caller-provided buffer (vunet_conv2d: amax_out); the tensor is then TAGGED with that buffer (and its version counter:
# an in-place change invalidates the tag) and a later convolution that reads it alone skips its own pass over it.
# Max-pool passes the tag on (|max-pool(x)| <= max|x|, in both directions).  Buffers come from a per-stream arena that is
# zeroed once per training step (ops.prepacked) -- one memset instead of one per convolution.
_amax_arena = {}


def _new_amax_out(dev):
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx) if _raw_stream is not None else 0)
    a = _amax_arena.get(key)
    if a is None or a[1] >= a[0].shape[0]:
        a = _amax_arena[key] = [torch.zeros(128, 1024, device=dev, dtype=torch.float32), 0]
    a[1] += 1
    return a[0][a[1] - 1]


def _zero_scalar(dev, shape=(1,)):
    """A zeroed fp32 accumulator for a loss kernel (they add their workgroups' partial sums into it): a slice of this
    step's zeroed arena -- no fill launch of its own.  Each accumulator gets 64 bytes of a row."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (idx, _raw_stream(idx) if _raw_stream is not None else 0)
    z = _zero_rows.get(key)
    if z is None or z[1] + 16 > z[0].numel():
        z = _zero_rows[key] = [_new_amax_out(dev), 0]
    z[1] += 16
    return z[0][z[1] - 16:z[1] - 15].view(shape)


_zero_rows = {}
_amax_retired = []


def _reset_amax_arena():
    """Start a new step's arenas.  The previous step's stay allocated for one more step: their slices are read by
    kernels on OTHER streams than the one that allocated them (the pose encoder's side stream, the weight-gradient
    companions), which the caching allocator does not know about -- it could hand a freed 4 KB slice to a new kernel of
    the allocating stream while such a reader is still in flight.  Every step ends with all streams joined, so one step
    of grace is enough (and costs no record_stream call per launch)."""
    _amax_retired[:] = [list(_amax_arena.values())]
    _amax_arena.clear()
    _zero_rows.clear()


def _tag_amax(t, buf):
    t._vunet_amax = (buf, t._version)


def _tagged_amax(t):
    rec = getattr(t, "_vunet_amax", None)
    return rec[0] if rec is not None and rec[1] == t._version else None


def carry_amax_tag(src, dst):
    """``dst`` is an alias of ``src`` (a pass-through output): it holds the same values, so the same maxima (and it is
    as much the output of a ReLU layer as ``src`` is)."""
    tag = _tagged_amax(src)
    if tag is not None:
        _tag_amax(dst, tag)
    if _is_relu_out(src):
        _tag_relu_out(dst)


# ---- ReLU masks applied by the PRODUCER of a gradient.  A frozen conv + ReLU layer (the VGG19 stack) whose input x is
# itself the output of a conv + ReLU layer multiplies its data gradient by [x > 0] in the kernel epilogue (16-byte reads of
# x) -- the mask the layer below would otherwise apply while STAGING its dy tile (dword loads of a second tensor, 1.5x
# with the halo).  Max-pool and the L1 taps do the same for their inputs.  The gradient is tagged with the address of the
# ReLU output it was masked by; the layer below checks the tag and runs the plain data-gradient kernel.  A mask applied
# twice changes nothing, so correctness never rests on a tag -- a missing one only costs the old path.
_relu_premask = os.environ.get("VUNET_RELU_PREMASK", "1") != "0"


def enable_relu_premask(on: bool = True):
    global _relu_premask
    _relu_premask = bool(on)


def _tag_relu_out(t):
    t._vunet_relu_out = t._version


def _is_relu_out(t) -> bool:
    return t is not None and getattr(t, "_vunet_relu_out", None) == t._version


def _tag_masked(g, by):
    g._vunet_masked = (by.data_ptr(), g._version)


def _is_masked_by(g, y) -> bool:
    rec = getattr(g, "_vunet_masked", None)
    return rec is not None and y is not None and rec[0] == y.data_ptr() and rec[1] == g._version


def _amax_for(x1, x2=None):
    """The |x| partial maxima of a convolution's input ([x1: 512 | x2: 512]): the producers' tags where the sources
    carry one, a pass over the tensor (vunet_absmax_partials) where they do not."""
    t1 = _tagged_amax(x1)
    if x2 is None:
        return t1 if t1 is not None else absmax_partials(x1)
    t2 = _tagged_amax(x2)
    if t1 is None and t2 is None:
        return absmax_partials(x1, x2)
    # the two sources' maxima live in buffers of their own (a producer's tag / a pass over one source): handed to the
    # kernels as a PAIR (vunet_conv2d_a2 / vunet_conv2d_wgrad_a2) -- no launch to concatenate them
    return (t1 if t1 is not None else absmax_partials(x1), t2 if t2 is not None else absmax_partials(x2))


def _amax_pair(amax):
    """-> (amax, amax2) pointers' tensors for the *_a2 entry points."""
    return amax if isinstance(amax, tuple) else (amax, None)


def _publishes_amax(desc, has_aux: bool, has_res: bool, has_wx: bool) -> bool:
    """Will vunet_conv2d leave the |y| maxima in amax_out for this problem (fp16 scheme)?"""
    key = ("p", desc.N, desc.C1, desc.C2, desc.Hs, desc.Ws, desc.M, desc.m_off, desc.Mpad, desc.Ho, desc.Wo, desc.KH,
           desc.stride, desc.pad, desc.mode, desc.in_act, desc.drop_p > 0, desc.out_act, desc.d2s, desc.aux_act,
           desc.aux_drop_p > 0, has_aux, has_res, has_wx)
    r = _wants_split_cache.get(key)
    if r is None:
        r = _wants_split_cache[key] = _lib.lib().vunet_conv2d_publishes_amax(ctypes.byref(desc), int(has_aux),
                                                                             int(has_res), 2 if has_wx else 0) == 1
    return r


def _wgrad_wants_split(wd) -> bool:
    key = ("w", wd.N, wd.C1, wd.C2, wd.Hs, wd.Ws, wd.Cout, wd.Ho, wd.Wo, wd.KH, wd.stride, wd.pad, wd.flags & 1)
    r = _wants_split_cache.get(key)
    if r is None:
        r = _wants_split_cache[key] = _lib.lib().vunet_conv2d_wgrad_wants_split(ctypes.byref(wd)) == 1
    return r


def _wants_split(desc, has_aux: bool, has_res: bool, has_mask: bool = False) -> bool:
    """Would vunet_conv2d route this problem to a split kernel (so that the h2 scheme owes it the |x| maxima)?"""
    key = (desc.N, desc.C1, desc.C2, desc.Hs, desc.Ws, desc.M, desc.m_off, desc.Mpad, desc.Ho, desc.Wo, desc.KH, desc.stride,
           desc.pad, desc.mode, desc.in_act, desc.drop_p > 0, desc.out_act, desc.d2s, desc.aux_act, desc.aux_drop_p > 0,
           has_aux, has_res, has_mask, _scheme())
    r = _wants_split_cache.get(key)
    if r is None:
        r = _wants_split_cache[key] = _lib.lib().vunet_conv2d_wants_split(ctypes.byref(desc), int(has_aux), int(has_res),
                                                                          int(has_mask), _scheme()) == 1
    return r


def x6_mtiles(m: int) -> int:
    """vunet_x6_mtiles: 32-channel tiles of a split weight image's M dimension (padded)."""
    return (((m + 31) // 32 + 1) + 1) & ~1


def x6_image_units(cout: int, c1: int, c2: int, k: int, dgrad: bool, split: int = 1) -> int:
    """16-byte units of a layer's split weight image (vunet_x6_image_bytes / 16); 0: geometry not covered.
    ``split`` 1: three bf16 terms (576 units per slab), 2: two fp16 terms (384 per slab + the header unit)."""
    if k != 3:
        return 0
    slab, hdr = (384, 1) if split == 2 else (576, 0)
    if dgrad:
        return (cout // 16) * 3 * x6_mtiles(c1 + c2) * slab + hdr if cout % 16 == 0 else 0
    return ((c1 + c2) // 16) * 3 * x6_mtiles(cout) * slab + hdr if (c1 % 16 == 0 and c2 % 16 == 0) else 0


def _alloc_x6(cout, c1, c2, k, dgrad, dev):
    if _scheme() == 0:
        return None
    n = x6_image_units(cout, c1, c2, k, dgrad, _scheme())
    # (zero-filled ONCE: the tiled pack never touches the padding m-tiles of an image, vunet_weightnorm_fwd_multi)
    return torch.zeros(n * 4, device=dev, dtype=torch.int32) if n else None


def pack_weights(v, g, bias, gamma, beta, c1: int, c2: int, kind: int, need_dgrad: bool):
    """vunet_weightnorm_fwd -> (wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d)."""
    cout, ctot, kh, kw = v.shape
    assert ctot == c1 + c2
    t = kh * kw
    dev = v.device
    wt_f = torch.empty(t * (_r2(c1) + _r2(c2)), _r32(cout), device=dev, dtype=torch.float32)
    wt_d = torch.empty(t * _r2(cout), _r32(ctot), device=dev, dtype=torch.float32) if need_dgrad else None
    wx_f = _alloc_x6(cout, c1, c2, kh, False, dev) if kh == kw else None
    wx_d = _alloc_x6(cout, c1, c2, kh, True, dev) if (need_dgrad and kh == kw) else None
    small = torch.empty(4, cout, device=dev, dtype=torch.float32)   # scale, shift, invnorm, row maxima of |w_eff|
    d = WnDesc(cout, c1, c2, kh, kw, kind, _scheme())
    _call("vunet_weightnorm_fwd", ctypes.byref(d), _p(v), _p(g), _p(bias), _p(gamma), _p(beta), _p(wt_f), _p(wt_d),
          _p(wx_f), _p(wx_d), _p(small[0]), _p(small[1]), _p(small[2]), _p(small[3]), _stream())
    return wt_f, wt_d, small[0], small[1], small[2], wx_f, wx_d


# ------------------------------------------------------------------------------------------------
# optional per-launch timing of the conv kernel families (bench.py roofline): HIP events on the launch stream
# ------------------------------------------------------------------------------------------------
_prof = {"on": False, "recs": []}
_NO_TIMER = contextlib.nullcontext()   # (the timer's key tuple -- a dozen ctypes field reads -- is built only when profiling)


def profile_start():
    _prof["on"], _prof["recs"] = True, []


def profile_stop(detail: bool = False, by_kernel: bool = False):
    """-> {family: {"ms": total kernel time, "flop": algorithmic FLOPs, "n": launches}} (synchronises the device)."""
    _prof["on"] = False
    if _prof["recs"]:
        torch.cuda.synchronize()
    fam = {}
    for name, flop, e0, e1 in _prof["recs"]:
        if by_kernel:
            name = name[-1]
        elif not detail:
            name = name[0]
        f = fam.setdefault(name, {"ms": 0.0, "flop": 0.0, "n": 0})
        f["ms"] += e0.elapsed_time(e1)
        f["flop"] += flop
        f["n"] += 1
    _prof["recs"] = []
    return fam


class _Timed:
    def __init__(self, name, flop):
        self.name, self.flop = name, flop

    def __enter__(self):
        if _prof["on"]:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if _prof["on"]:
            self.e1.record()
            _prof["recs"].append((self.name, self.flop, self.e0, self.e1))


def _conv_gather(desc: ConvDesc, x1, x2, wt, shift, res, aux, y, wx=None, amax=None, res2=None):
    """``amax``: the |x| partial maxima if the caller already has them (h2 scheme), else taken from the producer's tag or
    computed here when needed.  Returns the maxima the launch used (None if it did not need any).  When the fp16 kernel
    runs (and does not store through depth-to-space) y is tagged with the maxima of what it wrote."""
    t = desc.KH * desc.KW
    scheme = _scheme() if wx is not None else 0
    amax_out = None
    if scheme == 2:
        if _wants_split(desc, aux is not None, res is not None):
            if amax is None:
                amax = _amax_for(x1, x2)
        else:
            wx = amax = None
    else:
        amax = None
    if _scheme() == 2 and _publishes_amax(desc, aux is not None, res is not None, wx is not None):
        amax_out = _new_amax_out(y.device)   # the h2, 1x1 and LDS-tiled kernels leave max|y| there
    if desc.mode == 0:
        name, flop = "conv_gather_fwd", 2.0 * desc.N * desc.Ho * desc.Wo * desc.M * (desc.C1 + desc.C2) * t
    else:  # transposed gather: MACs of the forward conv restricted to this source
        name, flop = "conv_gather_dgrad", 2.0 * desc.N * desc.Hs * desc.Ws * desc.C1 * desc.M * t
    kname = ""
    if _prof["on"]:
        buf = ctypes.create_string_buffer(96)
        _call("vunet_conv2d_variant", ctypes.byref(desc), 0 if aux is None else 1, 0 if wx is None else scheme, 0, buf, 96)
        kname = buf.value.decode()
    with (_Timed((name, desc.N, desc.C1, desc.C2, desc.Hs, desc.Ws, desc.M, desc.KH, desc.stride, desc.in_act, kname),
                flop) if _prof["on"] else _NO_TIMER):
        a1, a2 = _amax_pair(amax)
        rc = _lib.lib().vunet_conv2d_a2(ctypes.byref(desc), _p(x1), _p(x2), _p(wt), _p(wx), _p(shift), _p(res), _p(res2),
                                        _p(aux), _p(y), _p(a1), _p(a2), _p(amax_out), _stream())
        if rc == -3 and res2 is not None:
            # a kernel without the second residual slot (nothing was launched): one residual in the epilogue, the other by
            # an add over the result -- whose maxima are then not what the epilogue published
            _call("vunet_conv2d_a2", ctypes.byref(desc), _p(x1), _p(x2), _p(wt), _p(wx), _p(shift), _p(res), None, _p(aux),
                  _p(y), _p(a1), _p(a2), None, _stream())
            y.add_(res2)
            amax_out = None
        elif rc != 0:
            raise RuntimeError(f"vunet_conv2d_a2 failed with code {rc}")
    if amax_out is not None:
        _tag_amax(y, amax_out)
    return amax


# ------------------------------------------------------------------------------------------------
# Plain convolution primitives that are closed under differentiation (forward / data gradient / weight
# gradient each differentiate into the other two).  They back the *double-backward* path only (R1 penalty,
# models/synth_discriminator.py:244-256): FusedConv.backward, when called with create_graph=True, re-expresses
# its layer with these primitives and ordinary differentiable tensor ops and lets autograd differentiate that.
# ------------------------------------------------------------------------------------------------
def _plain_pack(w, need_dgrad):
    cout, cin = w.shape[0], w.shape[1]
    return pack_weights(w.contiguous(), None, None, None, None, cin, 0, 1, need_dgrad)   # (wt_f, wt_d, ..., wx_f, wx_d)


def _plain_desc(n, cin, hs, ws, m, ho, wo, k, stride, pad, mode, mpad):
    return ConvDesc(N=n, C1=cin, C2=0, Hs=hs, Ws=ws, M=m, m_off=0, Mpad=mpad, Ho=ho, Wo=wo, KH=k, KW=k,
                    stride=stride, pad=pad, mode=mode, in_act=ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0,
                    out_act=ACT_NONE, d2s=0, aux_act=ACT_NONE, aux_slope=0.0, aux_drop_p=0.0, aux_drop_seed=0)


class ConvPlainFwd(torch.autograd.Function):
    """y = conv2d(x, w, stride, pad), no bias."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        _dev(x, w)
        x, w = _c(x), _c(w)
        n, cin, hs, ws = x.shape
        cout, _, k, _ = w.shape
        ho, wo = conv_out_size(hs, k, stride, pad), conv_out_size(ws, k, stride, pad)
        pk = _plain_pack(w, False)
        wt_f = pk[0]
        y = torch.empty(n, cout, ho, wo, device=x.device, dtype=torch.float32)
        _conv_gather(_plain_desc(n, cin, hs, ws, cout, ho, wo, k, stride, pad, 0, wt_f.shape[1]), x, None, wt_f, None,
                     None, None, y, pk[5])
        ctx.save_for_backward(x, w)
        ctx.geo = (stride, pad)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad = ctx.geo
        gx = ConvPlainDgrad.apply(gy, w, x.shape[2], x.shape[3], stride, pad) if ctx.needs_input_grad[0] else None
        gw = ConvPlainWgrad.apply(x, gy, w.shape[2], stride, pad) if ctx.needs_input_grad[1] else None
        return gx, gw, None, None


class ConvPlainDgrad(torch.autograd.Function):
    """dx = conv2d_transpose(dy, w): the data gradient of ConvPlainFwd for an input of size hs x ws."""

    @staticmethod
    def forward(ctx, dy, w, hs, ws, stride, pad):
        _dev(dy, w)
        dy, w = _c(dy), _c(w)
        n, cout, ho, wo = dy.shape
        cin, k = w.shape[1], w.shape[2]
        pk = _plain_pack(w, True)
        wt_d = pk[1]
        dx = torch.empty(n, cin, hs, ws, device=dy.device, dtype=torch.float32)
        _conv_gather(_plain_desc(n, cout, ho, wo, cin, hs, ws, k, stride, pad, 1, wt_d.shape[1]), dy, None, wt_d, None,
                     None, None, dx, pk[6])
        ctx.save_for_backward(dy, w)
        ctx.geo = (stride, pad)
        return dx

    @staticmethod
    def backward(ctx, gdx):
        dy, w = ctx.saved_tensors
        stride, pad = ctx.geo
        gdy = ConvPlainFwd.apply(gdx, w, stride, pad) if ctx.needs_input_grad[0] else None
        gw = ConvPlainWgrad.apply(gdx, dy, w.shape[2], stride, pad) if ctx.needs_input_grad[1] else None
        return gdy, gw, None, None, None, None


class ConvPlainWgrad(torch.autograd.Function):
    """dW[co, ci, kh, kw] = sum_px dy[co, px] x[ci, px (+) tap]: the weight gradient of ConvPlainFwd."""

    @staticmethod
    def forward(ctx, x, dy, k, stride, pad):
        _dev(x, dy)
        x, dy = _c(x), _c(dy)
        n, cin, hs, ws = x.shape
        cout, ho, wo = dy.shape[1], dy.shape[2], dy.shape[3]
        wd = WgradDesc(N=n, C1=cin, C2=0, Hs=hs, Ws=ws, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=stride, pad=pad,
                       in_act=ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0, nsplit=1,
                       flags=0 if _scheme() else 1)
        ns = _lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
        wd.nsplit = ns
        ktot = k * k * cin
        slabs = torch.empty(ns * _r32(cout) * (ktot + 1), device=x.device, dtype=torch.float32)
        dshift = slabs[ns * _r32(cout) * ktot:]
        _call("vunet_conv2d_wgrad", ctypes.byref(wd), _p(x), None, _p(dy), _p(slabs), _p(dshift), None, None, _stream())
        dw = torch.empty(cout, cin, k, k, device=x.device, dtype=torch.float32)
        work = torch.empty(cout * (ktot + 1), device=x.device, dtype=torch.float32)
        wn = WnDesc(cout, cin, 0, k, k, 1, 0)
        _call("vunet_weightnorm_bwd", ctypes.byref(wn), _p(slabs), _p(dshift), ns, _p(dw), None, None, None, None,
              _p(dw), None, None, None, None, _p(work), 0, _stream())
        ctx.save_for_backward(x, dy)
        ctx.geo = (k, stride, pad)
        return dw

    @staticmethod
    def backward(ctx, gw):
        x, dy = ctx.saved_tensors
        k, stride, pad = ctx.geo
        gx = ConvPlainDgrad.apply(dy, gw, x.shape[2], x.shape[3], stride, pad) if ctx.needs_input_grad[0] else None
        gdy = ConvPlainFwd.apply(x, gw, stride, pad) if ctx.needs_input_grad[1] else None
        return gx, gdy, None, None, None


def _differentiable_layer(x1, x2, res, v, g, bias, gamma, beta, cfg):
    """The layer of FusedConv.forward re-expressed with ConvPlainFwd and differentiable tensor ops."""
    x = x1 if x2 is None else torch.cat([x1, x2], dim=1)
    if cfg.in_act == ACT_ELU:
        x = torch.nn.functional.elu(x)
    elif cfg.in_act == ACT_RELU:
        x = torch.relu(x)
    elif cfg.in_act == ACT_LRELU:
        x = torch.nn.functional.leaky_relu(x, cfg.in_slope)
    if cfg.drop_p > 0:
        m1 = dropout_keep_mask(x1.shape, cfg.drop_p, cfg.drop_seed, x1.device)
        mask = m1 if x2 is None else torch.cat(
            [m1, dropout_keep_mask(x2.shape, cfg.drop_p, (cfg.drop_seed + SEED2_OFFSET) & 0xFFFFFFFF, x2.device)], dim=1)
        x = x * mask / (1.0 - cfg.drop_p)
    cout = v.shape[0]
    if cfg.kind == 1:
        w, scale_out = v, None
    else:
        nrm = v.flatten(1).norm(dim=1)
        if cfg.kind == 2:
            nrm = nrm.clamp_min(1e-12)
        w = v * ((g.reshape(-1) if cfg.kind == 0 else 1.0) / nrm).view(-1, 1, 1, 1)
        scale_out = gamma.reshape(1, cout, 1, 1) if gamma is not None else None
    y = ConvPlainFwd.apply(x, w, cfg.stride, cfg.pad)
    if bias is not None:
        y = y + bias.view(1, cout, 1, 1)
    if scale_out is not None:
        y = scale_out * y
    if cfg.kind != 1 and beta is not None:
        y = y + beta.reshape(1, cout, 1, 1)
    if cfg.out_act == ACT_RELU:
        y = torch.relu(y)
    elif cfg.out_act == ACT_SIGMOID:
        y = torch.sigmoid(y)
    elif cfg.out_act == ACT_ELU:
        y = torch.nn.functional.elu(y)
    elif cfg.out_act == ACT_LRELU:
        y = torch.nn.functional.leaky_relu(y, cfg.in_slope)
    if cfg.d2s:
        y = DepthToSpace.apply(y)
    if res is not None:
        y = y + res
    return y


# Parameters whose .grad is a view of a flat bucket (optim.FlatBucket) are written in place by the
# weight-norm backward kernel (accumulate mode); autograd then gets None for them.  Listeners (the
# data-parallel averager) are told which parameter gradients have just been completed.
_grad_hooks = []
_direct_grads_enabled = True
_wgrad_streams = {"on": False, "by_stream": {}, "dirty": set()}


def enable_wgrad_streams(on: bool = True):
    """Run the weight-gradient kernels of backward on companion HIP streams (one per stream that runs backward
    nodes).  Only layers whose parameter gradients are written in place into flat buckets take part; the caller
    must call ``join_wgrad_streams()`` after ``backward()`` and before reading those gradients."""
    _wgrad_streams["on"] = bool(on)


_stream_objs = {}


def _current_stream_obj():
    """torch.cuda.current_stream() builds a Stream object per call (2.7 us); the objects are cached by raw handle."""
    if _raw_stream is None:
        return torch.cuda.current_stream()
    idx = torch.cuda.current_device()
    key = (idx, _raw_stream(idx))
    st = _stream_objs.get(key)
    if st is None:
        st = _stream_objs[key] = torch.cuda.current_stream()
    return st


def _wgrad_stream_for(cur):
    ws = _wgrad_streams["by_stream"].get(cur.cuda_stream)
    if ws is None:
        ws = _wgrad_streams["by_stream"][cur.cuda_stream] = torch.cuda.Stream()
    _wgrad_streams["dirty"].add(cur.cuda_stream)
    return ws


def join_wgrad_streams():
    """The current stream waits for every companion stream that has been handed work since the last join (only those:
    a stream that took no part in a hipGraph capture must not be waited on from inside it)."""
    flush_weight_grads()
    cur = torch.cuda.current_stream()
    for key in _wgrad_streams["dirty"]:
        ws = _wgrad_streams["by_stream"][key]
        # (the data-parallel hooks call this from INSIDE a companion stream's flush: a stream never waits on itself -- a
        # no-op when launched eagerly, a node that depends on itself when the step is being captured into a hipGraph)
        if ws.cuda_stream != cur.cuda_stream:
            cur.wait_stream(ws)
    _wgrad_streams["dirty"].clear()


_inference_bf16 = False


class inference_precision:
    """``with ops.inference_precision("bf16"):`` -- under ``torch.no_grad()`` the 3x3 convolutions the bf16 kernel
    covers round their operands to bf16 and accumulate in fp32 (vunet_conv2d_bf16); everything else, and every call
    that records a graph, stays fp32.  This is the precision BASELINE config 5 names for the render loop."""

    def __init__(self, dtype: str = "bf16"):
        if dtype not in ("bf16", "f32", "fp32"):
            raise ValueError(f"unknown inference precision {dtype!r}")
        self.on = dtype == "bf16"

    def __enter__(self):
        global _inference_bf16
        self.prev, _inference_bf16 = _inference_bf16, self.on

    def __exit__(self, *exc):
        global _inference_bf16
        _inference_bf16 = self.prev
        return False


class direct_grads:
    """``with ops.direct_grads(False):`` around ``torch.autograd.grad`` calls that ask for parameter gradients:
    those must come back as tensors, not be accumulated into ``.grad`` (models/synth_discriminator.py:197-205)."""

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)

    def __enter__(self):
        global _direct_grads_enabled
        self.prev, _direct_grads_enabled = _direct_grads_enabled, self.enabled

    def __exit__(self, *exc):
        global _direct_grads_enabled
        _direct_grads_enabled = self.prev
        return False


def _has_direct_grad(p) -> bool:
    return (_direct_grads_enabled and getattr(p, "_vunet_direct_grad", False) and p.grad is not None
            and p.grad.is_contiguous())


def add_grad_hook(fn):
    _grad_hooks.append(fn)


def remove_grad_hook(fn):
    if fn in _grad_hooks:
        _grad_hooks.remove(fn)


# ------------------------------------------------------------------------------------------------
# batched weight preparation: every layer of a model in two launches (vunet_weightnorm_fwd_multi)
# ------------------------------------------------------------------------------------------------
class WnItem(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("v", "g", "bias", "gamma", "beta", "wt_f", "wt_d", "scale", "shift",
                                                "invnorm", "wx_f", "wx_d", "wmax")] + [("d", WnDesc)]


_prepack_sets = {}      # id(model) -> dict(signature, table, max_cout, entries, keep)
_active_prepack = {}    # id(module) -> (split, wt_f, wt_d, scale, shift, invnorm), only inside `prepacked(...)`


def _ptr_or_none(t):
    return None if t is None else t.data_ptr()


def _build_prepack_set(model):
    convs = model.__dict__.get("_vunet_conv_modules")   # the conv layers of the model (its structure is static)
    if convs is None:
        convs = [m for m in model.modules() if hasattr(m, "_params")]
        object.__setattr__(model, "_vunet_conv_modules", convs)
    mods = [m for m in convs if m.__dict__.get("_last_split") is not None]
    sig = (_scheme(),) + tuple((id(m), m.__dict__["_last_split"]) for m in mods)
    cur = _prepack_sets.get(id(model))
    if cur is not None and cur["signature"] == sig:
        return cur
    if not mods:
        return None
    dev = next(model.parameters()).device
    items, entries, keep, max_cout = (WnItem * len(mods))(), {}, [], 1
    for i, m in enumerate(mods):
        v, g, b, gamma, beta = m._params()
        c1, c2, need_x = m._last_split
        cout, ctot, kh, kw = v.shape
        t = kh * kw
        # (zero-filled once: the tiled pack writes what its tiles own -- columns past Ctot of wt_d stay as allocated)
        wt_f = torch.zeros(t * (_r2(c1) + _r2(c2)), _r32(cout), device=dev, dtype=torch.float32)
        wt_d = torch.zeros(t * _r2(cout), _r32(ctot), device=dev, dtype=torch.float32) if need_x else None
        wx_f = _alloc_x6(cout, c1, c2, kh, False, dev) if kh == kw else None
        wx_d = _alloc_x6(cout, c1, c2, kh, True, dev) if (need_x and kh == kw) else None
        small = torch.empty(4, cout, device=dev, dtype=torch.float32)
        it = items[i]
        it.v, it.g, it.bias, it.gamma, it.beta = (_ptr_or_none(x) for x in (v, g, b, gamma, beta))
        it.wt_f, it.wt_d = wt_f.data_ptr(), _ptr_or_none(wt_d)
        it.wx_f, it.wx_d = _ptr_or_none(wx_f), _ptr_or_none(wx_d)
        it.scale, it.shift, it.invnorm = small[0].data_ptr(), small[1].data_ptr(), small[2].data_ptr()
        it.wmax = small[3].data_ptr()
        it.d = WnDesc(cout, c1, c2, kh, kw, m.kind, _scheme())
        entries[id(m)] = ((c1, c2, need_x), wt_f, wt_d, small[0], small[1], small[2], wx_f, wx_d)
        keep.append((wt_f, wt_d, small, wx_f, wx_d))
        max_cout = max(max_cout, cout)
    raw = bytes(items)
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
    cur = {"signature": sig, "table": table, "n": len(mods), "max_cout": max_cout, "entries": entries, "keep": keep}
    _prepack_sets[id(model)] = cur
    return cur


def invalidate_active_prepack(module):
    """Parameters of ``module`` changed inside a ``prepacked`` block (L2NormConv2d's data-dependent init): its
    calls fall back to packing on the spot until the next block repacks."""
    _active_prepack.pop(id(module), None)


class prepacked:
    """``with ops.prepacked(model[, model2 ...]):`` -- pack the weights of every already-seen conv layer of the models in two
    launches per model and let the fused convs inside the block use them.  The caller guarantees that a model's parameters
    do not change between the block's start and the last use of that model inside it (the training step wraps forward +
    backward, the optimisers step after the last use: the generator's before the discriminator's own passes, which do not
    touch the generator; the discriminator's at the very end)."""

    _depth = 0

    def __init__(self, *models):
        self.models = [m for m in models if m is not None]
        self._keys = []

    def __enter__(self):
        # (nested blocks -- the regressor's five Adam steps inside the training step each re-fold the regressor -- add their
        # entries and take only those away again; the per-step arena of |y| maxima is the outermost block's)
        if prepacked._depth == 0:
            _reset_amax_arena()
        prepacked._depth += 1
        try:
            for model in self.models:
                cur = _build_prepack_set(model)
                if cur is not None:
                    _call("vunet_weightnorm_fwd_multi", _p(cur["table"]), cur["n"], cur["max_cout"], _stream())
                    _active_prepack.update(cur["entries"])
                    self._keys.extend(cur["entries"].keys())
        except BaseException:
            # __exit__ is not called when __enter__ raises: undo what this block added, or stale folded weights would stay
            # active across optimiser steps and every later outermost block would skip its reset
            self.__exit__(None, None, None)
            raise
        return self

    def __exit__(self, *a):
        prepacked._depth -= 1
        if prepacked._depth == 0:
            _active_prepack.clear()
        else:
            for k in self._keys:
                _active_prepack.pop(k, None)
        self._keys = []


# ------------------------------------------------------------------------------------------------
# deferred, batched weight-norm backward (vunet_weightnorm_bwd_multi).  A layer whose parameter gradients are all
# written in place into flat buckets only launches its weight-gradient kernel during backward; the slab reduction and
# the (dv, dg, dbias, dgamma, dbeta) arithmetic of up to _WN_GROUP such layers then run as ONE pair of launches on the
# stream that produced the slabs -- at the latest when backward ends (autograd engine callback) or when
# ``flush_weight_grads`` / ``join_wgrad_streams`` is called.  Slabs and workspaces are per-layer persistent buffers, so
# the device-side item table of a group is identical from step to step and is uploaded once.
# ------------------------------------------------------------------------------------------------
_WN_GROUP = int(os.environ.get("VUNET_WN_GROUP", "64"))   # layers per batched launch pair, at most
_WN_ARENA_FLOATS = 48 << 20     # 192 MB of slabs per stream between two flushes: they stay in the 256 MB Infinity Cache
_wn_batch = {"on": os.environ.get("VUNET_WN_BATCH", "1") != "0", "pending": {}, "arena": {}, "tables": {},
             "callback_queued": False, "capture": os.environ.get("VUNET_WN_BATCH_CAPTURE", "1") != "0",
             "wgrad": os.environ.get("VUNET_WGRAD_BATCH", "1") != "0"}   # (A/B: the short weight-gradient launches batched too)
_wgrad_batchable_cache = {}


def _wgrad_batchable(wd) -> bool:
    # (everything the C-side predicate looks at; NOT the dropout seed, which changes from step to step when issued eagerly:
    #  keyed by the whole descriptor the cache grew by one entry per layer and step)
    key = (wd.N, wd.C1, wd.C2, wd.Hs, wd.Ws, wd.Cout, wd.Ho, wd.Wo, wd.KH, wd.KW, wd.stride, wd.pad, wd.in_act, wd.drop_p > 0,
           wd.nsplit, wd.flags)
    r = _wgrad_batchable_cache.get(key)
    if r is None:
        r = _wgrad_batchable_cache[key] = _lib.lib().vunet_conv2d_wgrad_batchable(ctypes.byref(wd)) == 1
    return r


def enable_wn_batching(on: bool = True):
    flush_weight_grads()
    _wn_batch["on"] = bool(on)


def _wn_buffers(n_slab: int, n_work: int, dev):
    """Slabs + workspace of one deferred layer: the next slice of the current stream's arena.  Backward visits the layers
    in the same order every step, so a layer gets the same addresses every step (the item tables stay valid); the arena
    restarts after every flush -- the flush kernels are ordered before the next writers on the same stream -- so a
    step's slab traffic recycles the same cache-resident memory instead of streaming gigabytes through HBM."""
    idx = torch.cuda.current_device()
    key = (idx, _raw_stream(idx))
    need = ((n_slab + 63) & ~63) + ((n_work + 63) & ~63)
    ar = _wn_batch["arena"].get(key)
    if ar is None or ar[0].numel() < need:
        flush_weight_grads()
        ar = _wn_batch["arena"][key] = [torch.empty(max(_WN_ARENA_FLOATS, need), device=dev, dtype=torch.float32), 0]
    if ar[1] + need > ar[0].numel():
        _wn_flush_stream(key)
    off = ar[1]
    ar[1] = off + need
    return ar[0][off:off + n_slab], ar[0][off + ((n_slab + 63) & ~63):off + ((n_slab + 63) & ~63) + n_work]


def _wn_flush_stream(stream_key):
    items = _wn_batch["pending"].pop(stream_key, None)
    ar = _wn_batch["arena"].get(stream_key)
    if ar is not None:
        ar[1] = 0
    if not items:
        return
    dev_idx, raw = stream_key
    sig = tuple(it[0] for it in items)
    tab = _wn_batch["tables"].get(sig)
    if tab is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("batched weight-norm backward: this group's item table has not been built yet and cannot be "
                               "uploaded while a hipGraph is being captured -- run an eager step of the same geometry on "
                               "the capture stream first (ShapePoseNet does)")
        arr = (WnBwdItem * len(items))()
        max_cout = max_blocks = 1
        for i, item in enumerate(items):
            fields = item[0]
            (slabs, dshift, v, g, bias, gamma, invnorm, dv, dg, dbias, dgamma, dbeta, work, dsc, ns) = fields
            it = arr[i]
            it.slabs, it.dshift, it.v, it.g, it.bias, it.gamma, it.invnorm = slabs, dshift, v, g, bias, gamma, invnorm
            it.dv, it.dg, it.dbias, it.dgamma, it.dbeta, it.workspace = dv, dg, dbias, dgamma, dbeta, work
            cout, c1, c2, kh, kw, kind = dsc
            it.d = WnDesc(cout, c1, c2, kh, kw, kind, 0)
            it.nsplit, it.accumulate = ns, 1
            max_cout = max(max_cout, cout)
            max_blocks = max(max_blocks, cout * ((kh * kw * (c1 + c2) + 63) // 64))
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(torch.device("cuda", dev_idx))
        tab = _wn_batch["tables"][sig] = (table, len(items), max_cout, max_blocks)
    table, n, max_cout, max_blocks = tab
    # layers whose weight-gradient LAUNCH was deferred too (short launches on small maps): all of them now, grouped by
    # kernel form -- a handful of launches instead of one per layer, ordered before the reduction that reads their slabs
    wg = [it[3] for it in items if len(it) > 3 and it[3] is not None]
    if wg:
        arr = (WgradItem * len(wg))()
        for i, (wd_bytes, ptrs) in enumerate(wg):
            ctypes.memmove(ctypes.addressof(arr[i].d), wd_bytes, ctypes.sizeof(WgradDesc))
            (arr[i].x1, arr[i].x2, arr[i].dy, arr[i].slabs, arr[i].dshift, arr[i].amax_x, arr[i].amax_x2,
             arr[i].amax_dy) = ptrs
        _call("vunet_conv2d_wgrad_multi", arr, len(wg), ctypes.c_void_p(raw))
    _call("vunet_weightnorm_bwd_multi", _p(table), n, max_cout, max_blocks, ctypes.c_void_p(raw))
    for it in items:
        for p_ in it[1]:
            for hook in _grad_hooks:
                hook(p_)


def flush_weight_grads():
    """Launch the deferred weight-norm backward of every stream that has pending layers (idempotent)."""
    for key in list(_wn_batch["pending"].keys()):
        _wn_flush_stream(key)
    _wn_batch["callback_queued"] = False


def _wn_defer(fields, params, keep, wgrad=None):
    """``keep``: tensors the item points at that nothing else is guaranteed to hold until the flush.  ``wgrad``: (descriptor
    bytes, pointers) of the layer's weight-gradient launch when that is deferred to the flush as well."""
    idx = torch.cuda.current_device()
    key = (idx, _raw_stream(idx))
    lst = _wn_batch["pending"].setdefault(key, [])
    lst.append((fields, params, keep, wgrad))
    if len(lst) >= _WN_GROUP:
        _wn_flush_stream(key)
    elif not _wn_batch["callback_queued"]:
        # the end of this backward pass flushes whatever is still pending (the engine runs the callback in the thread
        # that called backward(), after every node has run)
        _wn_batch["callback_queued"] = True
        torch.autograd.Variable._execution_engine.queue_callback(flush_weight_grads)


class FusedConv(torch.autograd.Function):
    """y = [d2s] act_out( conv( drop(act_in(cat(x1, x2))) ; w_eff ) + shift ) [+ res]."""

    @staticmethod
    def forward(ctx, x1, x2, res, v, g, bias, gamma, beta, cfg: ConvCfg):
        _dev(x1, x2, res, v, g, bias, gamma, beta)
        x1, x2, res = _c(x1), _c(x2), _c(res)
        v, g, bias, gamma, beta = _c(v), _c(g), _c(bias), _c(gamma), _c(beta)
        n, c1, hs, ws = x1.shape
        c2 = 0 if x2 is None else x2.shape[1]
        cout = v.shape[0]
        k = cfg.k
        ho, wo = conv_out_size(hs, k, cfg.stride, cfg.pad), conv_out_size(ws, k, cfg.stride, cfg.pad)
        need_x = (ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]))
        need_w = any(ctx.needs_input_grad[3:8])
        frozen = not any(t is not None and t.requires_grad for t in (v, g, bias, gamma, beta))
        # frozen feature extractor (VGG19): pack once, reuse for every pass.  The packed buffers hang off the weight
        # tensor itself (they die with it) and are stamped with every operand's storage address and version counter.
        sub = stamp = hit = pre = None
        if frozen:
            sub = (c1, c2, cfg.kind, bool(need_x), _scheme())
            stamp = tuple(None if t is None else (t.data_ptr(), t._version) for t in (v, g, bias, gamma, beta))
            hit = getattr(v, "_vunet_frozen_pack", {}).get(sub)
        if cfg.owner is not None and not frozen:
            old = cfg.owner.__dict__.get("_last_split")
            new = (c1, c2, bool(need_x) or (old is not None and old[2]))
            if new != old:
                object.__setattr__(cfg.owner, "_last_split", new)   # (nn.Module.__setattr__ costs ~2 us per call)
            pre = _active_prepack.get(id(cfg.owner))
            if pre is not None and (pre[0][:2] != (c1, c2) or (need_x and pre[2] is None)):
                pre = None
        if pre is not None:
            _, wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = pre
        elif hit is not None and hit[0] == stamp:
            wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = hit[1]
        else:
            wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d = pack_weights(v, g, bias, gamma, beta, c1, c2, cfg.kind,
                                                                          need_x)
            if frozen:
                if not hasattr(v, "_vunet_frozen_pack"):
                    v._vunet_frozen_pack = {}
                v._vunet_frozen_pack[sub] = (stamp, (wt_f, wt_d, scale, shift, invnorm, wx_f, wx_d))
        if cfg.d2s:
            y = torch.empty(n, cout // 4, 2 * ho, 2 * wo, device=x1.device, dtype=torch.float32)
        else:
            y = torch.empty(n, cout, ho, wo, device=x1.device, dtype=torch.float32)
        d = ConvDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=ws, M=cout, m_off=0, Mpad=wt_f.shape[1], Ho=ho, Wo=wo, KH=k, KW=k,
                     stride=cfg.stride, pad=cfg.pad, mode=0, in_act=cfg.in_act, in_slope=cfg.in_slope,
                     drop_p=cfg.drop_p, drop_seed=cfg.drop_seed, out_act=cfg.out_act, d2s=int(cfg.d2s))
        ctx.amax_x = None
        if cfg.bf16 and _lib.lib().vunet_conv2d_bf16_supported(ctypes.byref(d)) == 1:
            # render path (models/vunets.py:508-515 under no_grad): bf16 operands, fp32 accumulate
            wb = torch.empty((c1 + c2) * 9 * wt_f.shape[1], device=x1.device, dtype=torch.bfloat16)
            _call("vunet_pack_bf16", _p(wt_f), _p(wb), c1, c2, wt_f.shape[1], _stream())
            with (_Timed(("conv_bf16_fwd", n, c1, c2, hs, ws, cout, k, cfg.stride, cfg.in_act, "conv_bf16_kernel"),
                        2.0 * n * ho * wo * cout * (c1 + c2) * k * k) if _prof["on"] else _NO_TIMER):
                _call("vunet_conv2d_bf16", ctypes.byref(d), _p(x1), _p(x2), _p(wb), _p(shift), _p(res), _p(y), _stream())
        else:
            ctx.amax_x = _conv_gather(d, x1, x2, wt_f, shift, res, None, y, wx_f)   # reused by the weight gradient
        ctx.cfg = cfg
        # packed by ops.prepacked: invnorm lives in the model's persistent pack set (the batched weight-norm backward keeps
        # raw pointers to it in a table that is uploaded once)
        ctx.prepacked = pre is not None
        ctx.param_refs = (v, g, bias, gamma, beta)  # the caller's tensors (Parameters): direct .grad writes
        ctx.res_ref = res
        ctx.dims = (n, c1, c2, hs, ws, cout, ho, wo)
        ctx.need_w = need_w
        ctx.save_for_backward(x1, x2, v, g, bias, gamma, invnorm, wt_d,
                              y if cfg.out_act in (ACT_SIGMOID, ACT_RELU, ACT_LRELU, ACT_ELU) else None, wx_d)
        if cfg.out_act == ACT_RELU:
            _tag_relu_out(y)
        ctx.x1_relu_out = _relu_premask and _is_relu_out(x1)
        if cfg.passthrough:
            # second output: x1 itself (autograd makes it an alias with this node as grad_fn).  Whoever else reads the
            # tensor reads the alias, so that gradient arrives HERE (g_alias) and is added in the data-gradient
            # kernel's epilogue -- no aten::add_ over the full tensor when autograd would have summed the two.
            ctx.set_materialize_grads(False)
            return y, x1
        return y

    @staticmethod
    def backward(ctx, dy, g_alias=None):
        x1, x2, v, g, bias, gamma, invnorm, wt_d, y, wx_d = ctx.saved_tensors
        cfg: ConvCfg = ctx.cfg
        if dy is None:   # (pass-through layers only: the convolution's own output was not used)
            return g_alias, None, None, None, None, None, None, None, None
        if g_alias is not None:
            g_alias = _c(g_alias)
        if torch.is_grad_enabled():
            # create_graph=True (e.g. the R1 penalty): build a differentiable backward by re-expressing the layer
            # with primitives that autograd can differentiate again, and differentiating that expression
            res, beta = ctx.res_ref, ctx.param_refs[4]
            with torch.enable_grad():
                y2 = _differentiable_layer(x1, x2, res, v, g, bias, gamma, beta, cfg)
            needs = list(ctx.needs_input_grad[:8])
            if cfg.res_is_x1:
                needs[2] = False   # the residual IS source 1: its gradient is already part of d/dx1
            tensors = (x1, x2, res, v, g, bias, gamma, beta)
            wanted = [t for t, need in zip(tensors, needs) if need and t is not None]
            grads = iter(torch.autograd.grad(y2, wanted, dy, create_graph=True, allow_unused=True))
            out = [next(grads) if (need and t is not None) else None for t, need in zip(tensors, needs)]
            if g_alias is not None:
                out[0] = g_alias if out[0] is None else out[0] + g_alias
            return (*out, None)
        n, c1, c2, hs, ws, cout, ho, wo = ctx.dims
        dy = _c(dy)
        dres = dy if (ctx.needs_input_grad[2] and not cfg.res_is_x1) else None
        k = cfg.k
        # frozen ReLU layers (the VGG19 stack): the ReLU backward rides in the data-gradient kernel's staging
        premasked = cfg.out_act == ACT_RELU and _is_masked_by(dy, y)   # the producer of dy already applied [y > 0]
        if (cfg.out_act == ACT_RELU and not premasked and not ctx.need_w and not cfg.d2s and x2 is None
                and ctx.needs_input_grad[0] and cfg.in_act == ACT_NONE and cfg.drop_p == 0 and wt_d is not None):
            dx = torch.empty_like(x1)
            d = ConvDesc(N=n, C1=cout, C2=0, Hs=ho, Ws=wo, M=c1, m_off=0, Mpad=wt_d.shape[1], Ho=hs, Wo=ws, KH=k, KW=k,
                         stride=cfg.stride, pad=cfg.pad, mode=1, in_act=ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0,
                         out_act=ACT_NONE, d2s=0)
            kname = ""
            if _prof["on"]:
                buf = ctypes.create_string_buffer(96)
                _call("vunet_conv2d_variant", ctypes.byref(d), 0, 0 if wx_d is None else _scheme(), 1, buf, 96)
                kname = buf.value.decode()
                if kname.startswith("conv_tiled"):
                    kname = kname.replace(", 1, 0, ", ", 1, 4, ")
            with (_Timed(("conv_gather_dgrad", n, cout, 0, ho, wo, c1, k, cfg.stride, 0, kname),
                        2.0 * n * ho * wo * cout * c1 * k * k) if _prof["on"] else _NO_TIMER):
                rc = -3
                if wx_d is not None and _wants_split(d, False, cfg.res_is_x1, True):
                    amax = _amax_for(dy) if _scheme() == 2 else None
                    amax_out = _new_amax_out(dx.device) if _scheme() == 2 else None
                    rc = _lib.lib().vunet_conv2d_dgrad_relu_x6(ctypes.byref(d), _p(dy), _p(y), _p(wx_d),
                                                               _p(dy if cfg.res_is_x1 else g_alias), _p(dx), _p(amax),
                                                               _p(amax_out), _stream())
                    if rc == 0 and amax_out is not None:
                        _tag_amax(dx, amax_out)
                if rc == -3:
                    rc = _lib.lib().vunet_conv2d_dgrad_relu(ctypes.byref(d), _p(dy), _p(y), _p(wt_d),
                                                            _p(dy if cfg.res_is_x1 else g_alias), _p(dx), _stream())
            if rc == 0:
                if cfg.res_is_x1 and g_alias is not None:
                    dx.add_(g_alias)
                return dx, None, dres, None, None, None, None, None, None
            if rc != -3:   # anything but VUNET_ERR_UNSUPPORTED is an error; unsupported geometries take the two-pass route
                raise RuntimeError(f"vunet_conv2d_dgrad_relu failed with code {rc}")
            if _prof["on"] and _prof["recs"]:
                _prof["recs"].pop()
        dconv = dy
        if cfg.out_act != ACT_NONE and not premasked:
            dconv = torch.empty_like(dy)
            _call("vunet_act_bwd_from_out", _p(y), _p(dy), _p(dconv), cfg.out_act, cfg.in_slope, dy.numel(), _stream())
        if cfg.d2s:
            t = torch.empty(n, cout, ho, wo, device=dy.device, dtype=torch.float32)
            _call("vunet_space_to_depth", _p(dconv), _p(t), n, cout // 4, 2 * ho, 2 * wo, _stream())
            carry_amax_tag(dconv, t)   # a permutation: the same values, the same maxima
            dconv = t
        dv = dg = dbias = dgamma = dbeta = None
        dy_amax = []   # |dconv| maxima of the h2 scheme: computed once (on this stream), shared by dgrad and wgrad

        def get_dy_amax():
            if not dy_amax:
                dy_amax.append(_amax_for(dconv))
            return dy_amax[0]

        wd = WgradDesc(N=n, C1=c1, C2=c2, Hs=hs, Ws=ws, Cout=cout, Ho=ho, Wo=wo, KH=k, KW=k, stride=cfg.stride,
                       pad=cfg.pad, in_act=cfg.in_act, in_slope=cfg.in_slope, drop_p=cfg.drop_p,
                       drop_seed=cfg.drop_seed, nsplit=1, flags=(0, 0, 2)[_scheme()] if _scheme() else 1)
        wg_amax = (None, None)
        if ctx.need_w and _scheme() == 2 and _wgrad_wants_split(wd):
            # (the forward may not have needed the |x| maxima -- 1x1 / fp32 kernels -- but the producers' tags are still there)
            wg_amax = (ctx.amax_x if ctx.amax_x is not None else _amax_for(x1, x2), get_dy_amax())

        def weight_gradients(defer: bool = False):
            ns = _lib.lib().vunet_conv2d_wgrad_nsplit(ctypes.byref(wd))
            if ns < 1:
                raise RuntimeError(f"vunet_conv2d_wgrad_nsplit failed with code {ns}")
            wd.nsplit = ns
            ktot = k * k * (c1 + c2)
            n_slab = ns * _r32(cout) * ktot + ns * _r32(cout)
            if defer:   # per-layer persistent buffers: the batched reduction reads them after backward has moved on
                slabs, work = _wn_buffers(n_slab, cout * (ktot + 1), dy.device)
            else:
                slabs = torch.empty(n_slab, device=dy.device, dtype=torch.float32)
            dshift = slabs[ns * _r32(cout) * ktot:]
            kname = ""
            if _prof["on"]:
                buf = ctypes.create_string_buffer(96)
                _call("vunet_conv2d_wgrad_variant", ctypes.byref(wd), buf, 96)
                kname = buf.value.decode()
            ax1, ax2 = _amax_pair(wg_amax[0])
            # short launches (small maps, 1x1, the small stride-2 layers): the launch itself waits for the flush, where the
            # layers of one kernel form share a launch (vunet_conv2d_wgrad_multi); per-launch timing runs keep them apart
            late = (defer and _wn_batch["wgrad"] and not _prof["on"]
                    and _wgrad_batchable(wd))
            if not late:
                with (_Timed(("conv_wgrad", n, c1, c2, hs, ws, cout, k, cfg.stride, cfg.in_act, kname),
                            2.0 * n * ho * wo * cout * (c1 + c2) * k * k) if _prof["on"] else _NO_TIMER):
                    _call("vunet_conv2d_wgrad_a2", ctypes.byref(wd), _p(x1), _p(x2), _p(dconv), _p(slabs), _p(dshift),
                          _p(ax1), _p(ax2), _p(wg_amax[1]), _stream())
            ni = ctx.needs_input_grad
            if defer:   # every needed gradient is a view of a flat bucket: hand the layer to the batched backward
                outs = [p_.grad if (p_ is not None and need) else None for p_, need in zip(ctx.param_refs, ni[3:8])]
                ptrs = tuple(None if t is None else t.data_ptr() for t in
                             (slabs, dshift, v, g, bias, gamma, invnorm, outs[0], outs[1], outs[2], outs[3], outs[4], work))
                wg_item, keep = None, (invnorm,)
                if late:
                    wg_item = (bytes(wd), tuple(None if t is None else t.data_ptr() for t in
                                                (x1, x2, dconv, slabs, dshift, ax1, ax2, wg_amax[1])))
                    keep = (invnorm, x1, x2, dconv, ax1, ax2, wg_amax[1])
                _wn_defer(ptrs + ((cout, c1, c2, k, k, cfg.kind), ns),
                          [p_ for p_, o in zip(ctx.param_refs, outs) if o is not None], keep, wg_item)
                return None, None, None, None, None
            params = (v, g, bias, gamma, gamma)  # beta has gamma's shape
            outs, direct = [], []
            for idx, (p_, need) in enumerate(zip(ctx.param_refs, ni[3:8])):
                if p_ is None or not need:
                    outs.append(None)
                    direct.append(False)
                elif _has_direct_grad(p_):
                    outs.append(p_.grad)      # written in place (+=) by the kernel: no autograd accumulation pass
                    direct.append(True)
                else:
                    outs.append(torch.empty_like(params[idx]))
                    direct.append(False)
            n_direct = sum(direct)
            n_needed = sum(o is not None for o in outs)
            wn = WnDesc(cout, c1, c2, k, k, cfg.kind, 0)
            work = torch.empty(cout * (ktot + 1), device=dy.device, dtype=torch.float32)

            def run(sel_direct: bool):
                sel = [o if (o is not None and d_ == sel_direct) else None for o, d_ in zip(outs, direct)]
                _call("vunet_weightnorm_bwd", ctypes.byref(wn), _p(slabs), _p(dshift), ns, _p(v), _p(g), _p(bias),
                      _p(gamma), _p(invnorm), _p(sel[0]), _p(sel[1]), _p(sel[2]), _p(sel[3]), _p(sel[4]), _p(work),
                      1 if sel_direct else 0, _stream())
            if n_direct:
                run(True)
            if n_needed > n_direct:
                run(False)
            dv, dg, dbias, dgamma, dbeta = [None if d_ else o for o, d_ in zip(outs, direct)]
            for p_, d_ in zip(ctx.param_refs, direct):
                if d_:
                    for hook in _grad_hooks:
                        hook(p_)
            return dv, dg, dbias, dgamma, dbeta

        if ctx.need_w:
            ni_ = ctx.needs_input_grad
            all_direct = all((not ni_[3 + i]) or p_ is None or _has_direct_grad(p_) for i, p_ in enumerate(ctx.param_refs))
            # batched weight-norm backward.  While a hipGraph is being captured its item tables must already exist (they are
            # uploaded from host memory on first use): the trainer runs its eager steps on the capture stream first, so
            # that arenas, companion streams and tables are the captured step's own (_wn_flush_stream raises otherwise)
            defer_ok = (_wn_batch["on"] and ctx.prepacked and all_direct and dy.is_cuda
                        and (_wn_batch["capture"] or not torch.cuda.is_current_stream_capturing()))
            if _wgrad_streams["on"] and all_direct and dy.is_cuda:
                # weight gradients are off the critical path of backward (only the data gradient feeds the next
                # layer): run wgrad + slab reduce + weight-norm backward on a companion stream; the results land in
                # the flat gradient buckets, ordered before the optimiser by ops.join_wgrad_streams()
                cur = _current_stream_obj()
                wstream = _wgrad_stream_for(cur)
                wstream.wait_stream(cur)
                # tensors the caching allocator may hand out again before the companion stream has read them: the
                # activations and dy (freed as backward moves on), the |x| maxima if they are not arena slices, invnorm if
                # it is not part of the model's persistent pack set (the parameters themselves outlive the step)
                for t_ in (x1, x2, dconv, *_amax_pair(wg_amax[0]), wg_amax[1], None if ctx.prepacked else invnorm):
                    if t_ is not None:
                        t_.record_stream(wstream)
                torch.cuda.set_stream(wstream)   # (the context-manager form costs 6 us per layer, this pair 0.8)
                try:
                    weight_gradients(defer_ok)
                finally:
                    torch.cuda.set_stream(cur)
            else:
                dv, dg, dbias, dgamma, dbeta = weight_gradients(defer_ok)
        dx1 = dx2 = None
        if ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1]):
            def dgrad(x, cs, m_off, seed, add, mask_by_x=False, add2=None):
                dx = torch.empty_like(x)
                has_aux = cfg.in_act != ACT_NONE or cfg.drop_p > 0
                # x is a ReLU output and this layer has no prologue of its own: the epilogue's act'(aux) slot applies
                # [x > 0] for the layer below (see _relu_premask)
                mask_by_x = mask_by_x and not has_aux
                d = ConvDesc(N=n, C1=cout, C2=0, Hs=ho, Ws=wo, M=cs, m_off=m_off, Mpad=wt_d.shape[1], Ho=hs, Wo=ws,
                             KH=k, KW=k, stride=cfg.stride, pad=cfg.pad, mode=1, in_act=ACT_NONE, in_slope=0.0,
                             drop_p=0.0, drop_seed=0, out_act=ACT_NONE, d2s=0,
                             aux_act=ACT_RELU if mask_by_x else cfg.in_act,
                             aux_slope=cfg.in_slope, aux_drop_p=cfg.drop_p, aux_drop_seed=seed)
                has_aux = has_aux or mask_by_x
                amax = None
                if wx_d is not None and _scheme() == 2 and _wants_split(d, has_aux, add is not None):
                    amax = get_dy_amax()
                _conv_gather(d, dconv, None, wt_d, None, add, x if has_aux else None, dx, wx_d, amax, res2=add2)
                if mask_by_x:
                    _tag_masked(dx, x)
                return dx
            if ctx.needs_input_grad[0]:
                # (x is the residual AND has another reader: dy in the epilogue's first slot, that reader's gradient in its second)
                dx1 = dgrad(x1, c1, 0, cfg.drop_seed, dy if cfg.res_is_x1 else g_alias, ctx.x1_relu_out,
                            add2=g_alias if cfg.res_is_x1 else None)
                g_alias = None
            if x2 is not None and ctx.needs_input_grad[1]:
                dx2 = dgrad(x2, c2, c1, (cfg.drop_seed + SEED2_OFFSET) & 0xFFFFFFFF, None)
        if g_alias is not None:   # x1 itself needed no gradient from this layer
            dx1 = g_alias
        return dx1, dx2, dres, dv, dg, dbias, dgamma, dbeta, None


def fused_conv(x1, x2, res, v, g, bias, gamma, beta, cfg: ConvCfg):
    # backward treats the saved output as the activation's output: a residual added after an output activation
    # would corrupt it (no model on the path does this)
    assert res is None or cfg.out_act == ACT_NONE, "fused_conv: residual together with an output activation"
    if res is not None and res is x1 and not cfg.d2s and cfg.out_act == ACT_NONE:
        cfg.res_is_x1 = True
    cfg.bf16 = _inference_bf16 and not torch.is_grad_enabled()   # grad mode of the CALLER (forward() never records)
    if cfg.passthrough and not (torch.is_grad_enabled() and x1.requires_grad and x1.is_contiguous() and _grad_passthrough):
        cfg.passthrough = False
        return FusedConv.apply(x1, x2, res, v, g, bias, gamma, beta, cfg), x1
    if cfg.passthrough:
        y, alias = FusedConv.apply(x1, x2, res, v, g, bias, gamma, beta, cfg)
        carry_amax_tag(x1, alias)
        return y, alias
    return FusedConv.apply(x1, x2, res, v, g, bias, gamma, beta, cfg)


_grad_passthrough = os.environ.get("VUNET_GRAD_PASSTHROUGH", "1") != "0"


def enable_grad_passthrough(on: bool = True):
    """Fan-out gradients of tensors read by a loss term / skip connection AND the next layer are added inside that
    layer's (or the loss's) backward kernel (``ConvCfg.passthrough``, ``L1MeanThrough``) instead of by autograd.  Off:
    the plain graph -- same values (tests/test_hip_training.py compares the two bit for bit)."""
    global _grad_passthrough
    _grad_passthrough = bool(on)


# ------------------------------------------------------------------------------------------------
# small differentiable ops
# ------------------------------------------------------------------------------------------------
class DepthToSpace(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c // 4, 2 * h, 2 * w, device=x.device, dtype=x.dtype)
        _call("vunet_depth_to_space", _p(x), _p(y), n, c, h, w, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        n, c, h, w = dy.shape
        dx = torch.empty(n, 4 * c, h // 2, w // 2, device=dy.device, dtype=dy.dtype)
        _call("vunet_space_to_depth", _p(dy), _p(dx), n, c, h, w, _stream())
        return dx


class SpaceToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty(n, 4 * c, h // 2, w // 2, device=x.device, dtype=x.dtype)
        _call("vunet_space_to_depth", _p(x), _p(y), n, c, h, w, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        n, c, h, w = dy.shape
        dx = torch.empty(n, c // 4, 2 * h, 2 * w, device=dy.device, dtype=dy.dtype)
        _call("vunet_depth_to_space", _p(dy), _p(dx), n, c, h, w, _stream())
        return dx


class CropWindow(torch.autograd.Function):
    """y = x[:, :, oy:oy+P, ox:ox+P] with (oy, ox) read from DEVICE memory (``off``: int32[2] on the GPU): the launch
    arguments are the same every step, so the adversarial term's random patch can be part of a captured hipGraph
    (vunet_crop_window; the backward scatters into a zeroed tensor)."""

    @staticmethod
    def forward(ctx, x, off, size):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c, size, size, device=x.device, dtype=x.dtype)
        _call("vunet_crop_window", _p(x), _p(y), n * c, h, w, size, _p(off), _stream())
        ctx.off, ctx.geo = off, (n, c, h, w, size)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w, size = ctx.geo
        dx = torch.zeros(n, c, h, w, device=dy.device, dtype=dy.dtype)
        _call("vunet_crop_window_bwd", _p(_c(dy)), _p(dx), n * c, h, w, size, _p(ctx.off), _stream())
        return dx, None, None


class UpsampleBilinear2x(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode="bilinear") (align_corners False): lib/modules.py:172-175."""

    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c, 2 * h, 2 * w, device=x.device, dtype=x.dtype)
        _call("vunet_upsample_bilinear2x_fwd", _p(x), _p(y), n * c, h, w, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        n, c, h2, w2 = dy.shape
        dx = torch.empty(n, c, h2 // 2, w2 // 2, device=dy.device, dtype=dy.dtype)
        _call("vunet_upsample_bilinear2x_bwd", _p(dy), _p(dx), n * c, h2 // 2, w2 // 2, _stream())
        return dx


class Reparam(torch.autograd.Function):
    """z = eps * exp(logstd) + mu   (models/vunets.py:594-597)."""

    @staticmethod
    def forward(ctx, mu, logstd, eps):
        _dev(mu, logstd, eps)
        mu, logstd, eps = _c(mu), _c(logstd), _c(eps)
        z = torch.empty_like(mu)
        if eps is None:   # the noise is drawn inside the kernel (vunet_unit_sample) and kept for the backward
            eps = torch.empty_like(mu)
            _call("vunet_unit_sample", _p(mu), _p(logstd), _p(z), _p(eps), mu.numel(), next_dropout_seed(), _stream())
        else:
            _call("vunet_reparam_fwd", _p(mu), _p(logstd), _p(eps), _p(z), mu.numel(), _stream())
        ctx.save_for_backward(logstd, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logstd, eps = ctx.saved_tensors
        dz = _c(dz)
        dmu, dls = torch.empty_like(dz), torch.empty_like(dz)
        _call("vunet_reparam_bwd", _p(dz), _p(logstd), _p(eps), _p(dmu), _p(dls), dz.numel(), _stream())
        return dmu, dls, None


# Posterior / prior noise drawn INSIDE the sampling kernel (UnitSample, Reparam with eps=None) instead of by torch.randn_like.
# Off by default: the reference's modules draw through torch.randn_like, and code written against them may rely on that
# (torch.manual_seed streams, a patched randn_like that replays recorded draws -- tests/test_hip_training.py does).  The
# trainers of this package switch it on around their own steps: one launch per draw, constant launch arguments.
_kernel_noise = {"on": False}


def kernel_noise_enabled() -> bool:
    return _kernel_noise["on"]


class kernel_noise:
    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _kernel_noise["on"] = _kernel_noise["on"], self.on
        return self

    def __exit__(self, *exc):
        _kernel_noise["on"] = self.prev
        return False


class FanOut(torch.autograd.Function):
    """``n`` aliases of ``x`` for ``n`` readers: their gradients meet in THIS node's backward and are summed by one launch
    that also publishes the sum's |.| maxima (vunet_sum_amax) -- instead of n - 1 ``aten::add_`` launches of the autograd
    engine and a maxima pass in front of the next data gradient.  For the small maps of the bottleneck, where every launch is
    latency.  The sum runs over the gradients in REVERSE reader order -- the order in which the engine, which runs later-created
    nodes first, would have accumulated them (same bits as the plain graph)."""

    @staticmethod
    def forward(ctx, x, n: int):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [_c(g) for g in reversed(gs) if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        _dev(*gs)
        out = torch.empty_like(gs[0])
        amax_out = _new_amax_out(out.device) if _scheme() == 2 else None
        ptrs = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
        _call("vunet_sum_amax", ptrs, len(gs), _p(out), _p(amax_out), out.numel(), _stream())
        if amax_out is not None:
            _tag_amax(out, amax_out)
        return out, None


_FAN_OUT = os.environ.get("VUNET_FAN_OUT", "1") != "0"   # (A/B switch)


def fan_out(x, n: int):
    """``n`` handles on ``x`` whose gradients are summed in one launch (FanOut) when the graph is recording on the GPU and
    gradient pass-through is on; otherwise ``x`` itself ``n`` times."""
    if n < 2 or not (_grad_passthrough and _FAN_OUT and x.is_cuda and torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    outs = FanOut.apply(x, n)
    for o in outs:
        carry_amax_tag(x, o)
    return outs


class UnitSample(torch.autograd.Function):
    """z = mu + eps with eps ~ N(0, 1) drawn inside the kernel (vunet_unit_sample; models/vunets.py:151-156): one launch
    instead of zeros_like + randn_like + the reparametrisation kernel, and no backward launch (dz/dmu = 1).  The seed
    follows the dropout seeds' sequence (next_dropout_seed), the step-to-step variation under a captured graph comes
    from the device step counter (set_dropout_step)."""

    @staticmethod
    def forward(ctx, mu):
        _dev(mu)
        mu = _c(mu)
        z = torch.empty_like(mu)
        _call("vunet_unit_sample", _p(mu), None, _p(z), None, mu.numel(), next_dropout_seed(), _stream())
        return z

    @staticmethod
    def backward(ctx, dz):
        return dz


def set_schedule_(lr_dev, lr: float, imax_dev, imax: float, step_dev, step: int):
    """This step's learning rate / information_max / dropout step into their device scalars: one launch (vunet_set_schedule)."""
    _call("vunet_set_schedule", _p(lr_dev), float(lr), _p(imax_dev), float(imax), _p(step_dev), int(step), _stream())


class TotalLoss(torch.autograd.Function):
    """(loss, likelihood_loss) of the training step from its scalar terms, ONE launch each way (vunet_total_loss / _bwd):
    likelihood_loss = ll_weight * sum(terms), loss = likelihood_loss + gamma * kl once the KL term is on -- the reference's
    torch.stack / sum / mul / add chain (experiments/shape_and_pose_net.py:391-405) and its five backward nodes.  ``gamma``: a
    device scalar (the controller's state, no gradient) or a Python float."""

    @staticmethod
    def forward(ctx, kl, gamma, ll_weight: float, use_kl: bool, *terms):
        _dev(kl, *terms)
        n = len(terms)
        g_t = gamma if torch.is_tensor(gamma) else None
        g_c = 0.0 if g_t is not None else float(gamma)
        loss = torch.empty((), device=kl.device, dtype=torch.float32)
        ll = torch.empty((), device=kl.device, dtype=torch.float32)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
        _call("vunet_total_loss", ptrs, n, _p(kl), _p(g_t), g_c, float(ll_weight), int(bool(use_kl)), _p(loss), _p(ll), _stream())
        ctx.cfg = (n, g_t, g_c, float(ll_weight), int(bool(use_kl)), [t.shape for t in terms], kl.shape)
        return loss, ll

    @staticmethod
    def backward(ctx, g_loss, g_ll):
        n, g_t, g_c, llw, use_kl, shapes, kl_shape = ctx.cfg
        ref = g_loss if g_loss is not None else g_ll
        d = torch.empty(n + 1, device=ref.device, dtype=torch.float32)
        _call("vunet_total_loss_bwd", _p(None if g_loss is None else _c(g_loss)), _p(None if g_ll is None else _c(g_ll)), n,
              _p(g_t), g_c, llw, use_kl, _p(d), _stream())
        return (d[n:n + 1].view(kl_shape) if use_kl else None, None, None, None,
                *[d[i:i + 1].view(sh) for i, sh in enumerate(shapes)])


def gamma_update_(gamma, imax, avg_kl, gamma_step: float):
    """gamma <- max(gamma - gamma_step * (information_max - avg_kl), 0) in place, all three device scalars: one launch
    (vunet_gamma_update; experiments/shape_and_pose_net.py:82-85, 442)."""
    _dev(gamma, imax, avg_kl)
    _call("vunet_gamma_update", _p(gamma), _p(imax), _p(avg_kl), float(gamma_step), _stream())
    return gamma


class L1Mean(torch.autograd.Function):
    """weight * mean|target - pred|, shape [1]; gradient flows to pred only (lib/losses.py:98-102)."""

    @staticmethod
    def forward(ctx, target, pred, weight: float):
        _dev(target, pred)
        target, pred = _c(target), _c(pred)
        out = _zero_scalar(pred.device)
        partial = torch.empty(1024, device=pred.device, dtype=torch.float32)
        _call("vunet_l1_mean_fwd", _p(target), _p(pred), _p(partial), _p(out), float(weight), pred.numel(), _stream())
        ctx.save_for_backward(target, pred)
        ctx.weight = float(weight)
        ctx.pred_relu_out = _relu_premask and _is_relu_out(pred)
        return out

    @staticmethod
    def backward(ctx, gout):
        target, pred = ctx.saved_tensors
        return None, _l1_backward(target, pred, None, ctx.weight, gout, ctx.pred_relu_out), None


def _l1_backward(target, pred, add, weight, gout, relu_mask=False):
    """db = add + weight/n * gout * sign(pred - target); under the fp16 scheme the kernel also leaves the partial maxima
    of |db| for the data gradient that reads it (one pass over the tensor less per VGG tap).  ``relu_mask``: ``pred`` is
    a ReLU layer's output -- db is zeroed where pred <= 0 and tagged so (see _relu_premask)."""
    db = torch.empty_like(pred)
    amax_out = _new_amax_out(db.device) if (_scheme() == 2 and db.is_cuda) else None
    # the upstream scalar gradient stays on the device (no host sync): the kernel reads gout[0]
    _call("vunet_l1_mean_bwd_amax", _p(target), _p(pred), _p(add), _p(db), weight / pred.numel(), _p(_c(gout)),
          pred.numel(), _p(amax_out), int(relu_mask), _stream())
    if amax_out is not None:
        _tag_amax(db, amax_out)
    if relu_mask:
        _tag_masked(db, pred)
    return db


class L1MeanThrough(torch.autograd.Function):
    """``L1Mean`` that also hands ``pred`` on: (loss, pred_alias).  A feature that feeds BOTH a loss term and the next
    layer then has ONE gradient consumer chain -- the next layer's gradient arrives here as ``g_alias`` and is added
    inside the L1 backward kernel, instead of autograd summing two full-size tensors (``aten::add_``, one extra pass
    over the largest activations of the step per VGG tap)."""

    @staticmethod
    def forward(ctx, target, pred, weight: float):
        _dev(target, pred)
        assert pred.is_contiguous()
        target = _c(target)
        out = _zero_scalar(pred.device)
        partial = torch.empty(1024, device=pred.device, dtype=torch.float32)
        _call("vunet_l1_mean_fwd", _p(target), _p(pred), _p(partial), _p(out), float(weight), pred.numel(), _stream())
        ctx.save_for_backward(target, pred)
        ctx.weight = float(weight)
        ctx.pred_relu_out = _relu_premask and _is_relu_out(pred)
        ctx.set_materialize_grads(False)
        return out, pred

    @staticmethod
    def backward(ctx, gout, g_alias):
        target, pred = ctx.saved_tensors
        if gout is None:
            return None, g_alias, None
        return None, _l1_backward(target, pred, None if g_alias is None else _c(g_alias), ctx.weight, gout,
                                  ctx.pred_relu_out), None


l1_pool_fusion = {"on": os.environ.get("VUNET_L1_POOL_FUSE", "1") != "0"}   # (A/B switch: tools/ab_step.py l1_pool)


class L1ThroughPool(torch.autograd.Function):
    """``L1Mean(target, pred)`` and ``MaxPool2(pred)`` as one node: (loss, pooled).  Both read ``pred`` (VGG19's relu1_2 /
    relu2_2 feed a loss term and the next pool); their two gradients -- the routed pool gradient and the L1 sign term --
    are formed in ONE pass over the tensor (vunet_l1_pool_bwd) instead of pool backward + L1 backward-with-add."""

    @staticmethod
    def forward(ctx, target, pred, weight: float):
        _dev(target, pred)
        target, pred = _c(target), _c(pred)
        n, c, h, w = pred.shape
        out = _zero_scalar(pred.device)
        partial = torch.empty(1024, device=pred.device, dtype=torch.float32)
        y = torch.empty(n, c, h // 2, w // 2, device=pred.device, dtype=pred.dtype)
        _call("vunet_l1_pool_fwd", _p(target), _p(pred), _p(partial), _p(out), _p(y), float(weight), n * c, h, w, _stream())
        tag = _tagged_amax(pred)
        if tag is not None:
            _tag_amax(y, tag)   # |max-pool(x)| <= max|x|
        ctx.save_for_backward(target, pred)
        ctx.weight = float(weight)
        ctx.pred_relu_out = _relu_premask and _is_relu_out(pred)
        ctx.set_materialize_grads(False)
        return out, y

    @staticmethod
    def backward(ctx, gout, gy):
        target, pred = ctx.saved_tensors
        n, c, h, w = pred.shape
        db = torch.empty_like(pred)
        amax_out = _new_amax_out(db.device) if _scheme() == 2 else None
        scale = ctx.weight / pred.numel() if gout is not None else 0.0
        _call("vunet_l1_pool_bwd", _p(target), _p(pred), _p(None if gy is None else _c(gy)), _p(db), scale,
              _p(None if gout is None else _c(gout)), n * c, h, w, _p(amax_out), int(ctx.pred_relu_out), _stream())
        if amax_out is not None:
            _tag_amax(db, amax_out)
        if ctx.pred_relu_out:
            _tag_masked(db, pred)
        return None, db, None


class KLPrior(torch.autograd.Function):
    """weight * kl_loss(mu, logstd) of lib/losses.py:283-291 (inputs [N, ...] flattened per sample)."""

    @staticmethod
    def forward(ctx, mu, logstd, weight: float):
        _dev(mu, logstd)
        mu, logstd = _c(mu), _c(logstd)
        n = mu.shape[0]
        d = mu.numel() // n
        out = _zero_scalar(mu.device, ())
        partial = torch.empty(256, device=mu.device, dtype=torch.float32)
        _call("vunet_kl_fwd", _p(mu), _p(logstd), _p(partial), _p(out), float(weight), n, d, _stream())
        ctx.save_for_backward(mu, logstd)
        ctx.scale = float(weight) / n
        return out

    @staticmethod
    def backward(ctx, gout):
        mu, logstd = ctx.saved_tensors
        dmu, dls = torch.empty_like(mu), torch.empty_like(mu)
        _call("vunet_kl_bwd", _p(mu), _p(logstd), _p(dmu), _p(dls), ctx.scale, _p(_c(gout)), mu.numel(), _stream())
        return dmu, dls, None


class SqDiff(torch.autograd.Function):
    """weight * mean_n sum_chw 0.5 (p-q)^2   (lib/losses.py:26-37)."""

    @staticmethod
    def forward(ctx, p, q, weight: float):
        _dev(p, q)
        p, q = _c(p), _c(q)
        n = p.shape[0]
        out = _zero_scalar(p.device, ())
        partial = torch.empty(256, device=p.device, dtype=torch.float32)
        _call("vunet_sqdiff_fwd", _p(p), _p(q), _p(partial), _p(out), float(weight), n, p.numel() // n, _stream())
        ctx.save_for_backward(p, q)
        ctx.scale = float(weight) / n
        return out

    @staticmethod
    def backward(ctx, gout):
        p, q = ctx.saved_tensors
        dp, dq = torch.empty_like(p), torch.empty_like(p)
        _call("vunet_sqdiff_bwd", _p(p), _p(q), _p(dp), _p(dq), ctx.scale, _p(_c(gout)), p.numel(), _stream())
        return dp, dq, None


class VggPreprocess(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        assert c == 3
        y = torch.empty_like(x)
        _call("vunet_vgg_preprocess", _p(x), _p(y), n, h, w, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        n, _, h, w = dy.shape
        dx = torch.empty_like(dy)
        _call("vunet_vgg_preprocess_bwd", _p(dy), None, _p(dx), n, h, w, _stream())
        return dx


class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty(n, c, h // 2, w // 2, device=x.device, dtype=x.dtype)
        _call("vunet_maxpool2_fwd", _p(x), _p(y), n * c, h, w, _stream())
        ctx.save_for_backward(x, y)
        ctx.x_relu_out = _relu_premask and _is_relu_out(x)
        tag = _tagged_amax(x)
        if tag is not None:
            _tag_amax(y, tag)   # |max-pool(x)| <= max|x|
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        dy = _c(dy)
        n, c, h, w = x.shape
        dx = torch.empty_like(x)
        _call("vunet_maxpool2_bwd_relu" if ctx.x_relu_out else "vunet_maxpool2_bwd", _p(x), _p(y), _p(dy), _p(dx), n * c, h,
              w, _stream())
        tag = _tagged_amax(dy)
        if tag is not None:
            _tag_amax(dx, tag)  # the gradient is routed (or zeroed), not scaled
        if ctx.x_relu_out:
            _tag_masked(dx, x)   # (see _relu_premask)
        return dx


class InstanceNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps: float):
        _dev(x)
        x = _c(x)
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        stats = torch.empty(n * c, 2, device=x.device, dtype=torch.float32)
        _call("vunet_instnorm_fwd", _p(x), _p(y), _p(stats), n * c, h * w, float(eps), _stream())
        ctx.save_for_backward(y, stats)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, stats = ctx.saved_tensors
        dy = _c(dy)
        n, c, h, w = y.shape
        dx = torch.empty_like(y)
        _call("vunet_instnorm_bwd", _p(y), _p(dy), _p(stats), _p(dx), n * c, h * w, _stream())
        return dx, None


class Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act: int, slope: float):
        _dev(x)
        x = _c(x)
        y = torch.empty_like(x)
        _call("vunet_act_fwd", _p(x), _p(y), act, float(slope), x.numel(), _stream())
        ctx.save_for_backward(y)
        ctx.act, ctx.slope = act, float(slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _call("vunet_act_bwd_from_out", _p(y), _p(dy), _p(dx), ctx.act, ctx.slope, dy.numel(), _stream())
        return dx, None, None


def dropout_keep_mask(shape, p: float, seed: int, device) -> torch.Tensor:
    """The conv prologue's keep-mask, materialised (tests / debugging)."""
    m = torch.empty(shape, device=device, dtype=torch.float32)
    _call("vunet_dropout_mask", _p(m), m.numel(), float(p), int(seed) & 0xFFFFFFFF, _stream())
    return m


def adam_step_flat(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _dev(param, grad, exp_avg, exp_avg_sq)
    _call("vunet_adam_step", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr), float(beta1),
          float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), _stream())


def adam_step_flat_dev(param, grad, exp_avg, exp_avg_sq, lr_dev, beta1, beta2, eps, weight_decay, step_dev,
                       grad_scale=1.0):
    """``adam_step_flat`` with the learning rate (float64 [1]) and the step count (int64 [1]) read from device memory:
    no per-step launch argument (vunet_adam_step_dev)."""
    _dev(param, grad, exp_avg, exp_avg_sq)
    if not (lr_dev.is_cuda and lr_dev.dtype == torch.float64 and step_dev.is_cuda and step_dev.dtype == torch.int64):
        raise RuntimeError("adam_step_flat_dev: lr as a float64 and step as an int64 device tensor")
    _call("vunet_adam_step_dev", _p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(lr_dev),
          float(beta1), float(beta2), float(eps), float(weight_decay), _p(step_dev), float(grad_scale), _stream())


# ------------------------------------------------------------------------------------------------
# "p2": pre-split activations (csrc/conv_p2.hip).  A planes tensor of logical shape [N, C, H, W] is an fp16 tensor
# [2, N, C / 8, H + 2, W + 2, 8] (hi / lo planes, channel blocks of 8, one-pixel zero border) plus 128 int32 of meta
# (scale exponent, slots for the maximum).  The buffers are zero-filled ONCE and reused: no kernel writes the border.
# ------------------------------------------------------------------------------------------------
P2_META_INTS = 128


class P2Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "C", "H", "W", "M", "relu")]


class Planes:
    """A planes tensor: ``buf`` (fp16, zero border), ``meta`` (int32[128]), the logical ``shape`` (N, C, H, W)."""

    __slots__ = ("buf", "meta", "shape")

    def __init__(self, shape, device, meta=None):
        n, c, h, w = shape
        assert c % 8 == 0
        self.shape = (int(n), int(c), int(h), int(w))
        self.buf = torch.zeros(2, n, c // 8, h + 2, w + 2, 8, device=device, dtype=torch.float16)
        self.meta = meta if meta is not None else torch.zeros(P2_META_INTS, device=device, dtype=torch.int32)

    def to_nchw(self):
        n, c, h, w = self.shape
        y = torch.empty(n, c, h, w, device=self.buf.device, dtype=torch.float32)
        _call("vunet_p2_to_nchw", _p(self.buf), _p(self.meta), _p(y), n, c, h, w, _stream())
        return y


def p2_from_nchw(x, out: Planes = None, relu: bool = False, amax=None, zero_meta: bool = True) -> Planes:
    """fp32 NCHW -> planes (the scale from the tensor's true maximum: its amax tag, or a pass over it).  ``zero_meta`` False:
    the caller has zeroed ``out.meta`` already (one fill for all the metas of a pass)."""
    _dev(x)
    x = _c(x)
    if out is None:
        out = Planes(tuple(x.shape), x.device)
    if amax is None:
        amax = _tagged_amax(x)
    if amax is None:
        amax = absmax_partials(x)
    n, c, h, w = x.shape
    if zero_meta:
        out.meta.zero_()
    _call("vunet_p2_from_nchw", _p(x), _p(amax), min(int(amax.numel()), 512), int(relu), _p(out.buf), _p(out.meta), n, c, h, w,
          _stream())
    return out


class P2Weights:
    """Weight image + constants of one frozen 3x3 layer for vunet_p2_conv (forward and, on demand, data gradient)."""

    def __init__(self, weight, bias):
        self.weight, self.bias = weight, bias
        self.cout, self.cin = int(weight.shape[0]), int(weight.shape[1])
        self._img = {}

    def image(self, dgrad: bool):
        rec = self._img.get(dgrad)
        stamp = (self.weight.data_ptr(), self.weight._version, None if self.bias is None else self.bias._version)
        if rec is None or rec[0] != stamp:
            nbytes = _lib.lib().vunet_p2_weight_image_bytes(self.cout, self.cin, int(dgrad))
            if nbytes <= 0:
                raise RuntimeError("p2 weight image: geometry not covered")
            dev = self.weight.device
            img = torch.empty(nbytes // 2, device=dev, dtype=torch.float16)
            wk = torch.empty(4, device=dev, dtype=torch.float32)
            work = torch.empty(2 * max(self.cout, self.cin), device=dev, dtype=torch.float32)
            _call("vunet_p2_pack_weights", _p(_c(self.weight.detach())), _p(None if self.bias is None else _c(self.bias.detach())),
                  self.cout, self.cin, int(dgrad), _p(img), _p(wk), _p(work), _stream())
            rec = self._img[dgrad] = (stamp, img, wk)
        return rec[1], rec[2]


def p2_conv_supported(n, c, h, w, m) -> bool:
    d = P2Desc(n, c, h, w, m, 0)
    return _lib.lib().vunet_p2_conv_supported(ctypes.byref(d)) == 1


def p2_conv(x: Planes, wts: P2Weights, out: Planes, relu: bool = True, dgrad: bool = False, mask: Planes = None):
    """out = relu(conv(x) + bias) (forward) or conv^T(x) [* (mask != 0)] (data gradient).  ``out.meta``'s maximum slots
    must be zero (the caller zeroes the metas of a pass in one fill)."""
    n, c, h, w = x.shape
    m = wts.cin if dgrad else wts.cout
    assert c == (wts.cout if dgrad else wts.cin) and out.shape == (n, m, h, w)
    img, wk = wts.image(dgrad)
    d = P2Desc(n, c, h, w, m, int(relu and not dgrad))
    timer = _NO_TIMER
    if _prof["on"]:
        buf = ctypes.create_string_buffer(96)
        _call("vunet_p2_conv_variant", ctypes.byref(d), buf, 96)
        timer = _Timed(("conv_gather_dgrad" if dgrad else "conv_gather_fwd", n, c, 0, h, w, m, 3, 1, 0, buf.value.decode()),
                       2.0 * n * h * w * c * m * 9)
    with timer:
        _call("vunet_p2_conv", ctypes.byref(d), _p(x.buf), _p(x.meta), _p(img), _p(wk),
              _p(None if dgrad or wts.bias is None else wts.bias), _p(None if mask is None else mask.buf), _p(out.buf), _p(out.meta),
              _stream())
    return out


def p2_l1_fwd(t: Planes, p: Planes, weight: float):
    """weight * mean |t - p| of two planes tensors -> fp32 [1] (vunet_p2_l1_fwd)."""
    n, c, h, w = p.shape
    assert t.shape == p.shape
    out = _zero_scalar(p.buf.device)
    partial = torch.empty(1024, device=p.buf.device, dtype=torch.float32)
    _call("vunet_p2_l1_fwd", _p(t.buf), _p(t.meta), _p(p.buf), _p(p.meta), _p(partial), _p(out), float(weight), n, c, h, w, _stream())
    return out


def p2_pool_fwd(p: Planes, out: Planes, t: Planes = None, weight: float = 0.0):
    """out = maxpool2x2(p); with ``t`` also the L1 term weight * mean |t - p| from the same pass -> fp32 [1] (else None)."""
    n, c, h, w = p.shape
    assert out.shape == (n, c, h // 2, w // 2)
    loss = partial = None
    if t is not None:
        loss = _zero_scalar(p.buf.device)
        partial = torch.empty(1024, device=p.buf.device, dtype=torch.float32)
    _call("vunet_p2_pool_fwd", _p(None if t is None else t.buf), _p(None if t is None else t.meta), _p(p.buf), _p(p.meta), _p(partial),
          _p(loss), float(weight), _p(out.buf), _p(out.meta), n, c, h, w, _stream())
    return loss


def p2_l1_bwd(t: Planes, p: Planes, add, g: Planes, gscale: float, gout):
    """g = [add] + gscale * gout[0] * sign(p - t), zeroed where p == 0 (vunet_p2_l1_bwd)."""
    n, c, h, w = p.shape
    _call("vunet_p2_l1_bwd", _p(t.buf), _p(t.meta), _p(p.buf), _p(p.meta), _p(None if add is None else add.buf),
          _p(None if add is None else add.meta), _p(g.buf), _p(g.meta), float(gscale), _p(gout), n, c, h, w, _stream())
    return g


def p2_pool_bwd(p: Planes, dy: Planes, g: Planes, t: Planes = None, gscale: float = 0.0, gout=None):
    """g = route(dy) [+ gscale * gout[0] * sign(p - t)], zeroed where p == 0 (vunet_p2_pool_bwd)."""
    n, c, h, w = p.shape
    _call("vunet_p2_pool_bwd", _p(None if t is None else t.buf), _p(None if t is None else t.meta), _p(p.buf), _p(p.meta), _p(dy.buf),
          _p(dy.meta), _p(g.buf), _p(g.meta), float(gscale), _p(gout), n, c, h, w, _stream())
    return g
