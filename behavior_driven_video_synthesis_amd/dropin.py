"""Run the reference's own ``main.py`` -- UNCHANGED -- on the MI355X path.

    cd <checkout of CompVis/behavior-driven-video-synthesis>
    python -m behavior_driven_video_synthesis_amd.dropin main.py --config config/shape_and_pose_net.yaml --gpu 0

``install()`` registers this package's modules in ``sys.modules`` under the names the reference imports them by --
``lib.modules``, ``lib.losses``, ``models.vunets``, ``models.imagenet_pretrained``, ``models.synth_discriminator``,
``models.pose_behavior_rnn``, ``models.flow.blocks``, ``models.flow.simple_flow`` -- so
``experiments/shape_and_pose_net.py``'s ``from models.vunets import VunetAlter`` (:10), ``from lib.losses import vgg_loss``
and friends resolve to the HIP-backed classes without a single edited line.  ``UnsupervisedTransformer2`` and
``ResidualBehaviorNet`` train under autograd on this path (``experiments/behavior_net.py:591-714``: csrc/seq_train.hip,
csrc/seq_bptt.hip), so an unchanged ``main.py --config config/behavior_net.yaml`` runs both of its stages.  Each alias is a
HYBRID module: names this package defines come from here -- except ``HIDDEN`` ones, the flat-input building blocks of
``lib/modules.py`` (``ActNorm``, ``BasicFullyConnectedNet``) that the reference's OTHER flows (``ConditionalFlow``,
``SupervisedTransformer``, ... in its own ``models/flow/blocks.py``) import by name and use on 4-d inputs and under autograd:
those keep getting the reference's own classes; every other name (the flow / behaviour-net building blocks of ``lib/modules.py:236-707``,
the sequence losses of ``lib/losses.py``, ...) falls through to the reference's own file, loaded from the checkout under a
private name, so the experiments outside the hot path keep working.  ``lib`` / ``models`` stay the checkout's namespace
packages; only these sub-modules are replaced.

The reference's loop then drives these modules with its own ``torch.optim.Adam`` (autograd accumulates the gradients the
usual way; the flat-bucket fast path is ``experiments.shape_and_pose_net.ShapePoseNet`` of this package), and
``PerceptualVGG(vgg19(pretrained=True), ...)`` accepts torchvision's VGG19 as it stands (its conv weights are copied
into the fused conv + ReLU stack once).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import runpy
import sys
import types

ALIASES = {
    "lib.modules": "behavior_driven_video_synthesis_amd.lib.modules",
    "lib.losses": "behavior_driven_video_synthesis_amd.lib.losses",
    "models.vunets": "behavior_driven_video_synthesis_amd.models.vunets",
    "models.imagenet_pretrained": "behavior_driven_video_synthesis_amd.models.imagenet_pretrained",
    "models.synth_discriminator": "behavior_driven_video_synthesis_amd.models.synth_discriminator",
    # the behaviour front half of config 5 (experiments/behavior_net.py:14-17 imports these)
    "models.pose_behavior_rnn": "behavior_driven_video_synthesis_amd.models.pose_behavior_rnn",
    "models.flow.blocks": "behavior_driven_video_synthesis_amd.models.flow.blocks",
    "models.flow.simple_flow": "behavior_driven_video_synthesis_amd.models.flow.simple_flow",
}


# names this package defines for its own modules' use but does NOT export under the reference's module name
HIDDEN = {"lib.modules": ("ActNorm", "BasicFullyConnectedNet", "Linear", "_Marker")}


def _hybrid(alias: str, ours, ref_root: str):
    """Module ``alias`` whose attributes come from ``ours`` first, then from the reference's own file (loaded lazily)."""
    mod = types.ModuleType(alias, doc=f"{ours.__name__} (MI355X path) over the reference's {alias}")
    hidden = HIDDEN.get(alias, ())
    mod.__dict__.update({k: v for k, v in vars(ours).items() if not k.startswith("__") and k not in hidden})
    mod.__vunet_hip_alias__ = ours.__name__
    ref_file = os.path.join(ref_root, *alias.split(".")) + ".py"
    state = {"ref": None, "tried": False}

    def fallback(name):
        if name.startswith("__"):
            raise AttributeError(name)
        if not state["tried"]:
            state["tried"] = True
            if os.path.isfile(ref_file):
                spec = importlib.util.spec_from_file_location("_vunet_ref_" + alias.replace(".", "_"), ref_file)
                ref = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(ref)   # its own imports (lib.utils, ...) resolve inside the checkout as usual
                state["ref"] = ref
        if state["ref"] is not None and hasattr(state["ref"], name):
            return getattr(state["ref"], name)
        raise AttributeError(f"module {alias!r} has no attribute {name!r} (neither the MI355X package nor {ref_file})")

    mod.__getattr__ = fallback
    return mod


def install(ref_root: str = ".") -> dict:
    """Register the aliases; ``ref_root`` is the reference checkout (for the fall-through names).  Idempotent."""
    ref_root = os.path.abspath(ref_root)
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)      # what `python main.py` from the checkout does
    done = {}
    for alias, target in ALIASES.items():
        cur = sys.modules.get(alias)
        if cur is not None and getattr(cur, "__vunet_hip_alias__", None) == target:
            done[alias] = cur
            continue
        sys.modules[alias] = done[alias] = _hybrid(alias, importlib.import_module(target), ref_root)
    return done


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m behavior_driven_video_synthesis_amd.dropin main.py [the reference's arguments]")
    script = argv[0]
    install(os.path.dirname(os.path.abspath(script)) or ".")
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
