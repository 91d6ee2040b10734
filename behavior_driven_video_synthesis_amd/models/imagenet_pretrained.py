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

    def loss_terms(self, pred, target_features):
        """``{tap: w_tap * mean|target_tap - pred_tap|}`` of lib/losses.py:81-119 in ONE pass over ``pred``: each term is
        formed when its tap is computed and the next layer reads the alias ``ops.L1MeanThrough`` hands back, so a tap's
        two gradients (loss term, next layer) meet inside the L1 backward kernel instead of in an autograd add over
        the tensor.  Same values as ``forward`` + one ``L1Mean`` per tap."""
        weights = dict(zip(["input"] + [self.target_layers[k] for k in sorted(self.target_layers, key=int)],
                           self.loss_weights))
        losses = {}

        def term(name, t):
            losses[name], alias = ops.L1MeanThrough.apply(target_features[name], t, float(weights[name]))
            ops.carry_amax_tag(t, alias)
            return alias
        def term_pool(name, t):
            losses[name], pooled = ops.L1ThroughPool.apply(target_features[name], t, float(weights[name]))
            return pooled
        self._walk(pred, term, term_pool if ops.l1_pool_fusion["on"] else None)
        return losses
