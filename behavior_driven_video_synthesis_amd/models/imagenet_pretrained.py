"""MI355X-native counterpart of ``models/imagenet_pretrained.py:8-61`` (PerceptualVGG) plus the VGG19
``features`` container it wraps (torchvision.models.vgg19 is a third-party dependency of the reference,
absent here; its cfg-'E' topology is restated with torchvision's state-dict key names so a
user-supplied ``vgg19`` checkpoint loads unchanged).

Every conv + ReLU pair is one fused MFMA launch (ReLU epilogue); the input affine, the 2x2 max-pools
and their backward are HIP kernels.  The stack stops after module 31 (relu5_2): modules 32-36 of the
reference loop produce values that are discarded (models/imagenet_pretrained.py:53-61).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from .. import ops
from ..lib.modules import Conv2d

VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


class _Marker(nn.Module):
    """Parameter-free placeholder keeping torchvision's module indices (ReLU / MaxPool2d slots)."""

    def __init__(self, kind):
        super().__init__()
        self.kind = kind


class VGG19(nn.Module):
    """``features`` of torchvision's vgg19 (keys ``features.<idx>.{weight,bias}``), width-scalable for tests."""

    def __init__(self, width_div: int = 1):
        super().__init__()
        layers, cin = [], 3
        for v in VGG19_CFG:
            if v == "M":
                layers.append(_Marker("pool"))
            else:
                cout = max(v // width_div, 4)
                layers += [Conv2d(cin, cout, 3, padding=1), _Marker("relu")]
                cin = cout
        self.features = nn.Sequential(*layers)

    def init_synthetic(self, seed: int = 1234):
        """Seeded He-normal weights (no network access for the pretrained ones)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.features:
                if isinstance(m, Conv2d):
                    cout, cin = m.weight.shape[:2]
                    m.weight.copy_(torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9)))
                    m.bias.copy_((torch.rand(cout, generator=g) - 0.5) * 0.1)
        return self


def vgg19(pretrained: bool = False, weights_path: str = None, width_div: int = 1, seed: int = 1234,
          synthetic: bool = False) -> VGG19:
    """Stand-in for ``torchvision.models.vgg19(pretrained=True)`` (experiments/shape_and_pose_net.py:222).

    ``weights_path``: a torchvision vgg19 state dict (``features.N.weight``...; classifier keys ignored).
    Without it the weights are seeded-synthetic (stated as such by bench.py / DESIGN.md); asking for
    ``pretrained=True`` without a path warns unless the caller says ``synthetic=True`` -- a perceptual loss on a
    random VGG19 is a benchmark / test workload, not the reference's loss.
    """
    if pretrained and weights_path is None and not synthetic:
        import warnings
        warnings.warn("vgg19(pretrained=True) without weights_path: torchvision's weights cannot be fetched here, "
                      "using SEEDED-SYNTHETIC VGG19 weights (pass weights_path=<torchvision vgg19 state dict>, or "
                      "synthetic=True to silence this)", stacklevel=2)
    net = VGG19(width_div)
    if weights_path is not None:
        sd = torch.load(weights_path, map_location="cpu")
        net.load_state_dict({k: v for k, v in sd.items() if k.startswith("features.")})
    else:
        net.init_synthetic(seed)
    return net


def _adopt_features(features: nn.Sequential) -> nn.Sequential:
    """``features`` of this package's VGG19 pass through; a torchvision-style stack (``nn.Conv2d`` / ``nn.ReLU`` /
    ``nn.MaxPool2d`` at torchvision's indices -- what the reference hands over, experiments/shape_and_pose_net.py:222-229)
    is rebuilt ONCE as the fused conv + ReLU stack with the same weights, on the same device."""
    mods = list(features)
    if all(isinstance(m, (Conv2d, _Marker)) for m in mods):
        return features
    layers = []
    for m in mods:
        if isinstance(m, nn.Conv2d):
            if m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.bias is None:
                raise NotImplementedError("PerceptualVGG expects VGG-style 3x3 / stride 1 / pad 1 convolutions with bias")
            c = Conv2d(m.in_channels, m.out_channels, 3, padding=1)
            with torch.no_grad():
                c.weight.copy_(m.weight)
                c.bias.copy_(m.bias)
            layers.append(c.to(m.weight.device))
        elif isinstance(m, nn.ReLU):
            layers.append(_Marker("relu"))
        elif isinstance(m, nn.MaxPool2d):
            layers.append(_Marker("pool"))
        else:
            raise NotImplementedError(f"PerceptualVGG: unexpected module {type(m).__name__} in vgg.features")
    return nn.Sequential(*layers)


# ------------------------------------------------------------------------------------------------
# The stack on PRE-SPLIT activations ("p2", csrc/conv_p2.hip / planes.hip).  From relu1_1 on, every activation of the pass
# lives in HBM as two fp16 planes that ARE the matrix-core operand of the next layer (converted once, by the kernel that
# produces them); L1 taps, max-pools and the whole backward chain work on the planes directly.  Same arithmetic as the
# fp32-tensor path (two scaled fp16 terms per operand, three products, fp32 accumulation): tests/test_hip_p2.py.
# ------------------------------------------------------------------------------------------------
_p2_switch = {"on": __import__("os").environ.get("VUNET_VGG_P2", "1") != "0"}


def enable_p2(on: bool = True):
    """The perceptual loss's VGG19 pass on pre-split planes (default) or on fp32 NCHW tensors (the path every other layer uses)."""
    _p2_switch["on"] = bool(on)


class _P2Target(dict):
    """``PerceptualVGG.features_for_loss(target)`` on the p2 path: tap name -> the tap's planes BUFFER (a tensor, so that
    callers can ``record_stream`` the values), "input" -> the preprocessed image (fp32); ``planes``: name -> ops.Planes.
    Valid until the next target pass of the same PerceptualVGG (the buffers are persistent and reused)."""

    planes = None
    generation = 0


class _P2Engine:
    """Persistent state of the p2 pass of one PerceptualVGG and one input geometry: per role ("t": target pass, "p":
    prediction pass) the planes of every activation from relu1_1 to the last tap, for the prediction role also the
    gradient planes, and one meta arena per role that is zeroed by ONE fill at the start of a pass."""

    def __init__(self, program, n, h, w, device):
        self.program, self.geo, self.device = program, (n, h, w), device
        self.roles = {}
        self.generation = 0        # target passes so far (a _P2Target is valid until the next one)
        self.pred_generation = 0   # prediction passes so far (their activations live in ONE set of persistent buffers)

    @staticmethod
    def build_program(mods, last, taps):
        """mods: the (index, module) list of ``features``.  -> (first conv, [step...]) with step = ("conv", module, tap name
        or None) | ("pool",), or None if the stack is not conv+relu / pool shaped."""
        steps, first, i = [], None, 0
        while i < len(mods):
            name, m = mods[i]
            if int(name) > last:
                break
            if isinstance(m, Conv2d):
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                if nxt is None or not (isinstance(nxt[1], _Marker) and nxt[1].kind == "relu"):
                    return None
                tap = taps.get(nxt[0])
                if first is None:
                    if tap is not None:
                        return None
                    first = m
                else:
                    steps.append(("conv", m, tap))
                i += 2
            elif isinstance(m, _Marker) and m.kind == "pool":
                if first is None:
                    return None
                steps.append(("pool",))
                i += 1
            else:
                return None
        if first is None or not steps or steps[-1][0] != "conv" or steps[-1][2] is None:
            return None
        return first, steps

    def shapes(self):
        """(C, H, W) of A[0] .. A[L]; None if a layer is outside what vunet_p2_conv covers."""
        first, steps = self.program
        n, h, w = self.geo
        c = first.weight.shape[0]
        out = [(c, h, w)]
        for st in steps:
            if st[0] == "pool":
                if h % 2 or w % 2:
                    return None
                h, w = h // 2, w // 2
            else:
                m = st[1]
                if m.weight.shape[1] != c or tuple(m.weight.shape[2:]) != (3, 3) or not ops.p2_conv_supported(n, c, h, w, m.weight.shape[0]):
                    return None
                # the data gradient of the same layer
                if not ops.p2_conv_supported(n, m.weight.shape[0], h, w, c):
                    return None
                c = m.weight.shape[0]
            out.append((c, h, w))
        return out

    def _role(self, role):
        st = self.roles.get(role)
        if st is None:
            shp = self.shapes()
            n = self.geo[0]
            L = len(shp)
            metas = torch.zeros(3 * L, ops.P2_META_INTS, device=self.device, dtype=torch.int32)
            acts = [ops.Planes((n, c, h, w), self.device, meta=metas[k]) for k, (c, h, w) in enumerate(shp)]
            st = self.roles[role] = {"metas": metas, "A": acts, "G": [None] * L, "U": [None] * L, "shp": shp}
        return st

    def _grad_buf(self, st, kind, k):
        lst = st[kind]
        if lst[k] is None:
            c, h, w = st["shp"][k]
            L = len(st["shp"])
            lst[k] = ops.Planes((self.geo[0], c, h, w), self.device, meta=st["metas"][(1 if kind == "G" else 2) * L + k])
        return lst[k]

    @staticmethod
    def _weights(m):
        w = m.__dict__.get("_p2w")
        if w is None or w.weight is not m.weight:
            w = ops.P2Weights(m.weight, m.bias)
            object.__setattr__(m, "_p2w", w)
        return w

    def _first(self):
        """conv1_1 (3 input channels: the VALU kernel of csrc/conv_thin.hip, writing planes): its K-major weights for the
        forward and the data gradient and the bound constants, packed once (the stack is frozen)."""
        m = self.program[0]
        stamp = (m.weight.data_ptr(), m.weight._version, m.bias._version)
        rec = m.__dict__.get("_p2first")
        if rec is None or rec[0] != stamp:
            w, b = m.weight.detach().contiguous(), m.bias.detach().contiguous()
            wt_f, wt_d, _, shift, _, _, wx_d = ops.pack_weights(w, None, b, None, None, 3, 0, 1, True)
            wk = torch.empty(4, device=w.device, dtype=torch.float32)
            work = torch.empty(2 * w.shape[0], device=w.device, dtype=torch.float32)
            ops._call("vunet_p2_weight_bound", ops._p(w), ops._p(b), int(w.shape[0]), 3, ops._p(wk), ops._p(work), ops._stream())
            rec = (stamp, wt_f, wt_d, shift, wx_d, wk)
            object.__setattr__(m, "_p2first", rec)
        return rec

    def forward(self, role, xp, targets=None, tap_weights=None):
        """xp: the preprocessed image (fp32, 3 channels).  -> ({tap: Planes}, {tap: loss [1]})."""
        st = self._role(role)
        st["metas"].zero_()
        A = st["A"]
        _, wt_f, _, shift, _, wk = self._first()
        n, _, h, w = xp.shape
        amax = ops._tagged_amax(xp)
        if amax is None:
            amax = ops.absmax_partials(xp)
        ops._call("vunet_p2_conv_first", ops._p(xp), ops._p(amax), 512, ops._p(wt_f), int(wt_f.shape[1]), ops._p(shift), ops._p(wk),
                  ops._p(A[0].buf), ops._p(A[0].meta), n, h, w, A[0].shape[1], ops._stream())
        taps, losses = {}, {}
        steps = self.program[1]
        for k, step in enumerate(steps, start=1):
            if step[0] == "conv":
                ops.p2_conv(A[k - 1], self._weights(step[1]), A[k], relu=True)
                tap = step[2]
                if tap is not None:
                    taps[tap] = A[k]
                    pool_next = k < len(steps) and steps[k][0] == "pool"
                    if targets is not None and not pool_next:
                        losses[tap] = ops.p2_l1_fwd(targets[tap], A[k], tap_weights[tap])
            else:
                prev = steps[k - 2] if k >= 2 else None
                tap = prev[2] if (prev is not None and prev[0] == "conv") else None
                if targets is not None and tap is not None:
                    losses[tap] = ops.p2_pool_fwd(A[k - 1], A[k], t=targets[tap], weight=tap_weights[tap])
                else:
                    ops.p2_pool_fwd(A[k - 1], A[k])
        return taps, losses

    def backward(self, targets, tap_weights, gouts):
        """The data-gradient chain of the prediction pass, top tap down to relu1_1.  gouts: {tap: upstream gradient of the tap's
        loss term (device tensor [1]) or None}.  -> d loss / d relu1_1-output, ALREADY multiplied by [relu1_1 > 0], fp32."""
        st = self.roles["p"]
        A, steps = st["A"], self.program[1]
        L = len(steps)

        def tap_scale(tap, k):
            c, h, w = st["shp"][k]
            return float(tap_weights[tap]) / (self.geo[0] * c * h * w)

        def has(tap):
            return tap is not None and gouts.get(tap) is not None
        # D = gradient w.r.t. the convolution output under A[k] (i.e. after the ReLU backward), kept in G[k]
        k = L
        top = steps[L - 1][2]
        D = None
        if has(top):
            D = ops.p2_l1_bwd(targets[top], A[L], None, self._grad_buf(st, "G", L), tap_scale(top, L), gouts[top])
        while k >= 1:
            step = steps[k - 1]
            if step[0] == "pool":
                # D holds the gradient w.r.t. the pool OUTPUT A[k]; A[k-1] is a ReLU output, possibly a tap
                prev = steps[k - 2]
                tap = prev[2] if has(prev[2]) else None
                if D is None and tap is None:
                    k -= 1
                    continue
                if D is None:   # nothing from above: the tap's own step only
                    D = ops.p2_l1_bwd(targets[tap], A[k - 1], None, self._grad_buf(st, "G", k - 1), tap_scale(tap, k - 1), gouts[tap])
                else:
                    D = ops.p2_pool_bwd(A[k - 1], D, self._grad_buf(st, "G", k - 1), t=None if tap is None else targets[tap],
                                        gscale=0.0 if tap is None else tap_scale(tap, k - 1), gout=None if tap is None else gouts[tap])
                k -= 1
                continue
            # conv step k: A[k] = relu(conv(A[k-1])), D = gradient w.r.t. the conv output
            below = steps[k - 2] if k >= 2 else None          # what produced A[k-1]
            if D is None:
                tap = below[2] if (below is not None and below[0] == "conv" and has(below[2])) else None
                if tap is not None:
                    D = ops.p2_l1_bwd(targets[tap], A[k - 1], None, self._grad_buf(st, "G", k - 1), tap_scale(tap, k - 1), gouts[tap])
                k -= 1
                continue
            wts = self._weights(step[1])
            if below is None:                                   # A[0] = relu1_1: mask and leave the planes
                D = ops.p2_conv(D, wts, self._grad_buf(st, "G", 0), dgrad=True, mask=A[0])
            elif below[0] == "pool":                            # gradient w.r.t. a pool output: no ReLU of its own
                D = ops.p2_conv(D, wts, self._grad_buf(st, "G", k - 1), dgrad=True)
            elif has(below[2]):                                 # A[k-1] is a tap read by this layer too: add the tap's step, then mask
                up = ops.p2_conv(D, wts, self._grad_buf(st, "U", k - 1), dgrad=True)
                D = ops.p2_l1_bwd(targets[below[2]], A[k - 1], up, self._grad_buf(st, "G", k - 1), tap_scale(below[2], k - 1),
                                  gouts[below[2]])
            else:
                D = ops.p2_conv(D, wts, self._grad_buf(st, "G", k - 1), dgrad=True, mask=A[k - 1])
            k -= 1
        if D is None:
            return None
        # conv1_1's data gradient (64 -> 3 channels: VALU kernel) on the fp32 form of D
        _, _, wt_d, _, wx_d, _ = self._first()
        n, h, w = self.geo
        c0 = st["shp"][0][0]
        dconv = D.to_nchw()
        dx = torch.empty(n, 3, h, w, device=self.device, dtype=torch.float32)
        d = ops.ConvDesc(N=n, C1=c0, C2=0, Hs=h, Ws=w, M=3, m_off=0, Mpad=wt_d.shape[1], Ho=h, Wo=w, KH=3, KW=3, stride=1, pad=1,
                         mode=1, in_act=ops.ACT_NONE, in_slope=0.0, drop_p=0.0, drop_seed=0, out_act=ops.ACT_NONE, d2s=0)
        ops._conv_gather(d, dconv, None, wt_d, None, None, None, dx, wx_d)
        return dx


class P2VggLoss(torch.autograd.Function):
    """The prediction pass of the perceptual loss from conv1_1 up, on pre-split planes: (preprocessed image, engine, target)
    -> one loss term per tap.  ONE autograd node: the backward walks the data-gradient chain itself (conv / tap / pool
    kernels on planes, conv1_1's 64 -> 3 data gradient last) and hands back d loss / d preprocessed-image."""

    @staticmethod
    def forward(ctx, xp, engine, target, tap_weights, names):
        _, losses = engine.forward("p", xp, target.planes, tap_weights)
        engine.pred_generation += 1
        ctx.engine, ctx.target, ctx.tap_weights, ctx.names = engine, target, tap_weights, names
        ctx.generation, ctx.pred_generation = target.generation, engine.pred_generation
        return tuple(losses[n] for n in names)

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.target.generation != ctx.generation:
            raise RuntimeError("P2VggLoss.backward: the target features were overwritten by a later target pass of the same "
                               "PerceptualVGG (its planes buffers are persistent): run forward + backward per target")
        if ctx.engine.pred_generation != ctx.pred_generation:
            raise RuntimeError("P2VggLoss.backward: a later vgg_loss() of the same PerceptualVGG and input geometry has overwritten "
                               "this pass's activations (persistent planes buffers): call backward() before the next vgg_loss(), "
                               "or switch the planes path off (models.imagenet_pretrained.enable_p2(False)) for this use")
        g = ctx.engine.backward(ctx.target.planes, ctx.tap_weights,
                                {n: (None if go is None else go.contiguous()) for n, go in zip(ctx.names, gouts)})
        return g, None, None, None, None


class PerceptualVGG(nn.Module):
    """models/imagenet_pretrained.py:8-61: dict ``input, relu1_2, relu2_2, relu3_2, relu4_2, relu5_2``."""

    def __init__(self, vgg, weights):
        super().__init__()
        feats = vgg.module.features if isinstance(vgg, nn.DataParallel) else vgg.features
        self.vgg_layers = _adopt_features(feats)
        self.loss_weights = weights
        self.register_buffer("mean", torch.tensor([0.485, 0.456, 0.406], dtype=torch.float).view(1, 3, 1, 1))
        self.register_buffer("std", torch.tensor([0.229, 0.224, 0.225], dtype=torch.float).view(1, 3, 1, 1))
        self.target_layers = {"3": "relu1_2", "8": "relu2_2", "13": "relu3_2", "22": "relu4_2", "31": "relu5_2"}
        for p in self.vgg_layers.parameters():  # frozen feature extractor
            p.requires_grad_(False)

    def _walk(self, x, on_tap, on_tap_pool=None):
        """The feature pass, :41-61; ``on_tap(name, x) -> x`` sees every tapped feature before the next layer reads it;
        ``on_tap_pool(name, x) -> pooled`` (optional) takes a tap that is followed by a max-pool together with the pool."""
        x = on_tap("input", ops.VggPreprocess.apply(x))  # ((x+1)/2 - mean)/std, :43-44
        last = max(int(k) for k in self.target_layers)
        mods = list(self.vgg_layers._modules.items())
        i = 0
        while i < len(mods):
            name, m = mods[i]
            if int(name) > last:
                break
            if isinstance(m, Conv2d):
                nxt = mods[i + 1][1] if i + 1 < len(mods) else None
                if isinstance(nxt, _Marker) and nxt.kind == "relu":
                    x = m.fused(x, out_act=ops.ACT_RELU)  # conv + ReLU(inplace) as one launch
                    i += 1
                    name = mods[i][0]
                else:
                    x = m(x)
            elif m.kind == "pool":
                x = ops.MaxPool2.apply(x)
            else:
                x = ops.Activation.apply(x, ops.ACT_RELU, 0.0)
            if name in self.target_layers:
                nxt = mods[i + 1][1] if i + 1 < len(mods) else None
                if (on_tap_pool is not None and isinstance(nxt, _Marker) and nxt.kind == "pool"
                        and int(mods[i + 1][0]) <= last):
                    x = on_tap_pool(self.target_layers[name], x)   # the tap and the pool behind it as one node
                    i += 1
                else:
                    x = on_tap(self.target_layers[name], x)
            i += 1

    def forward(self, x):
        out = {}

        def keep(name, t):
            out[name] = t
            return t
        self._walk(x, keep)
        return out

    # ---- the pass on pre-split planes (see _P2Engine)
    def _p2_engine(self, x):
        """The engine for inputs shaped like ``x``, or None where the p2 kernels do not cover the stack / geometry."""
        if not (_p2_switch["on"] and x.is_cuda and x.dim() == 4 and ops.conv_precision() == "h2"):
            return None
        cache = self.__dict__.setdefault("_p2_engines", {})
        key = (tuple(x.shape), x.device.index)
        if key not in cache:
            eng = None
            prog = _P2Engine.build_program(list(self.vgg_layers._modules.items()), max(int(k) for k in self.target_layers),
                                           self.target_layers)
            if prog is not None and not any(p.requires_grad for p in self.vgg_layers.parameters()):
                eng = _P2Engine(prog, x.shape[0], x.shape[2], x.shape[3], x.device)
                first = prog[0]
                if (eng.shapes() is None or x.shape[1] != 3 or first.weight.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32
                        or first.weight.shape[0] % 8 or first.weight.shape[0] > 128):
                    eng = None
            cache[key] = eng
        return cache[key]

    def _tap_weights(self):
        names = ["input"] + [self.target_layers[k] for k in sorted(self.target_layers, key=int)]
        return dict(zip(names, (float(w) for w in self.loss_weights)))

    def features_for_loss(self, x):
        """What ``loss_terms`` / ``lib.losses.vgg_loss`` need of a TARGET image: on the p2 path the taps' planes (a
        ``_P2Target``; valid until the next call on this module), else ``forward(x)``."""
        eng = self._p2_engine(x)
        if eng is None or torch.is_grad_enabled():
            return self.forward(x)
        xp = ops.VggPreprocess.apply(x)
        taps, _ = eng.forward("t", xp)
        out = _P2Target({"input": xp})
        out.update({k: v.buf for k, v in taps.items()})
        out.planes = taps
        eng.generation += 1
        out.generation = eng.generation
        out.engine = eng
        return out

    def loss_terms(self, pred, target_features):
        """``{tap: w_tap * mean|target_tap - pred_tap|}`` of lib/losses.py:81-119 in ONE pass over ``pred``: each term is
        formed when its tap is computed and the next layer reads the alias ``ops.L1MeanThrough`` hands back, so a tap's
        two gradients (loss term, next layer) meet inside the L1 backward kernel instead of in an autograd add over
        the tensor.  Same values as ``forward`` + one ``L1Mean`` per tap."""
        weights = dict(zip(["input"] + [self.target_layers[k] for k in sorted(self.target_layers, key=int)],
                           self.loss_weights))
        losses = {}
        if isinstance(target_features, _P2Target):
            eng = target_features.engine
            if target_features.generation != eng.generation:
                raise RuntimeError("loss_terms: these target features were overwritten by a later features_for_loss() call")
            xp = ops.VggPreprocess.apply(pred)
            losses["input"], xp_alias = ops.L1MeanThrough.apply(target_features["input"], xp, float(weights["input"]))
            ops.carry_amax_tag(xp, xp_alias)
            names = tuple(st[2] for st in eng.program[1] if st[0] == "conv" and st[2] is not None)
            terms = P2VggLoss.apply(xp_alias, eng, target_features, self._tap_weights(), names)
            losses.update(dict(zip(names, terms)))
            return losses

        def term(name, t):
            losses[name], alias = ops.L1MeanThrough.apply(target_features[name], t, float(weights[name]))
            ops.carry_amax_tag(t, alias)
            return alias
        def term_pool(name, t):
            losses[name], pooled = ops.L1ThroughPool.apply(target_features[name], t, float(weights[name]))
            return pooled
        self._walk(pred, term, term_pool if ops.l1_pool_fusion["on"] else None)
        return losses
