"""Training-loop surface of the reference's ``experiments/shape_and_pose_net.py`` for the MI355X path.

Keeps what ``train_fn`` (:360-466), ``__update_gamma`` (:82-85), ``adjust_params`` (:500-512) and the
model / optimiser construction (:196-272) do, on synthetic or user-supplied batches, without the
ignite / wandb / dataset plumbing that is out of scope:

* ``VunetAlter`` + frozen ``PerceptualVGG`` + Adam(4 param groups, betas (0.5,0.9), lr linearly decayed),
* loss = ll_weight * sum_i vgg_loss_i + gamma * KL once ``iteration > n_init_batches``,
* gamma controller ``gamma <- max(gamma - gamma_step * (imax - kl), 0)``,
* optional regressor side loop (:407-425), which contributes no gradient to the VUnet,
* optional adversarial term (``training.gan``): the reference ships ``DiscTrainer`` / ``PartDiscriminator``
  (models/synth_discriminator.py:77-242) but its loop never constructs them; here they can be
  switched on -- generator loss ``weight * w * BCE(D(fake patch), 1)`` added to the step's loss, then one
  ``train_disc`` step on the same (detached) patches.  Off by default, so the default step is the reference's.

MI355X-first differences (results unchanged): the gamma controller and every logged scalar stay on
the device (the reference forces >= 10 host syncs per step), Adam is one fused launch per param
group, and under data parallelism the gradient buckets are all-reduced while backward still runs.
The checkpoint is ``{"model": state_dict, "optimizer": adam_state_dict}`` as at :474-482.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import ops
from ..lib.losses import compute_kl_with_prior, vgg_loss
from ..lib.utils import get_member, linear_var, n_parameters
from ..models.imagenet_pretrained import PerceptualVGG, vgg19
from ..models.synth_discriminator import DiscTrainer
from ..models.vunets import Regressor, VunetAlter
from ..optim import FusedAdam
from ..parallel import BucketedGradAverager, broadcast_parameters

DEFAULT_CONFIG = {
    # config/shape_and_pose_net.yaml (Human3.6m)
    "general": {"seed": 42, "debug": False},
    "data": {"dataset": "Human3.6m", "spatial_size": 256, "box_factor": 2, "bottleneck_factor": 2,
             "inplane_normalize": False},
    "architecture": {"n_latent_scales": 2, "conv_layer_type": "l1", "nf_start": 32, "nf_max": 128,
                     "subpixel_upsampling": True, "n_scales": 0, "n_rnb": 2, "linear_width_factor": 1,
                     "n_linear": 2, "cvae": False},
    "training": {"batch_size": 12, "vgg_weights": [1.0] * 6, "dropout_prob": 0.05, "lr": 0.0005,
                 "gamma_step": 0.00001, "n_init_batches": 4, "adam_betas": (0.5, 0.9), "end_iteration": 150000,
                 "imax_scaling": "none", "information_max": 1000, "ll_weight": 1.0, "train_regressor": True,
                 "weight_regressor": 4.0, "reg_steps": 5,
                 # adversarial term: not part of the reference loop, off by default
                 "gan": {"enabled": False, "weight": 1.0, "pd_scales": 3, "grad_pen": False, "lambda_gp": 10,
                         "grad_weighting": False, "lr": 0.0002, "save_intervall": 10000}},
}


class ShapePoseNet:
    def __init__(self, config: Dict, device="cuda:0", n_channels_x: int = 3, n_keypoints: int = 17,
                 vgg_weights_path: Optional[str] = None, vgg_width_div: int = 1, total_steps: Optional[int] = None,
                 process_group=None, vgg_synthetic: bool = False, vgg_seed: int = 1234,
                 hip_graph: Optional[bool] = None):
        self.config = config
        self.device = torch.device(device)
        arch, data, tr = config["architecture"], config["data"], config["training"]
        torch.manual_seed(config["general"].get("seed", 42))
        # ---- models (:196-235)
        kw = dict(arch)
        kw.update(data)
        kw["dropout_prob"] = tr.get("dropout_prob", 0.0)
        self.iteration = 0
        # :199-206: the l2 conv variant initialises gamma / beta from batch statistics during the first batches
        init_fn = (lambda: self.iteration <= tr["n_init_batches"]) if arch.get("conv_layer_type") == "l2" else None
        self.vunet = VunetAlter(init_fn=init_fn, n_channels_x=n_channels_x, **kw).to(self.device)
        overlap = self.device.type == "cuda" and bool(tr.get("two_streams", os.environ.get("VUNET_TWO_STREAMS", "1") != "0"))
        self.vunet.enable_two_streams(overlap)   # pose encoder (du) beside appearance encoder (eu, ed)
        ops.enable_wgrad_streams(overlap)        # weight gradients beside the data-gradient chain (process-wide switch)
        self.vgg = vgg19(pretrained=True, weights_path=vgg_weights_path, width_div=vgg_width_div, seed=vgg_seed,
                         synthetic=vgg_synthetic).to(self.device)
        self.vgg.eval()
        self.custom_vgg = PerceptualVGG(self.vgg, tr["vgg_weights"]).to(self.device)
        # ---- optimiser (:237-246)
        self.optimizer = FusedAdam(
            [{"params": list(get_member(self.vunet, n).parameters()), "name": n} for n in ("eu", "ed", "du", "dd")],
            lr=tr["lr"], betas=tuple(tr["adam_betas"]))
        self.train_regressor = bool(tr.get("train_regressor", False))
        if self.train_regressor:
            latent_widths = [data["spatial_size"] // (2 ** (self.vunet.n_scales - i))
                             for i in range(arch["n_latent_scales"], 0, -1)]
            self.regressor = Regressor(n_keypoints * 2, latent_widths=latent_widths, **arch).to(self.device)
            self.optimizer_regressor = FusedAdam(list(self.regressor.parameters()), lr=0.001)
        # ---- adversarial term (models/synth_discriminator.py:115-242), optional
        gan = tr.get("gan") or {}
        self.gan = None
        if gan.get("enabled", False):
            self.gan = DiscTrainer(self.vunet, {"pd_scales": gan.get("pd_scales", 3), "adam_beta": tuple(tr["adam_betas"]),
                                                "save_intervall": gan.get("save_intervall", 10000)},
                                   grad_pen=gan.get("grad_pen", False), lambda_gp=gan.get("lambda_gp", 10),
                                   grad_weighting=gan.get("grad_weighting", False), spatial_size=data["spatial_size"])
            self.gan.keep_generator_grads = True   # train_fn zeroes the flat gradient buckets itself
            self.gan.init_training([self.device], lr=gan.get("lr", tr["lr"]), process_group=process_group)
            self.gan_weight = float(gan.get("weight", 1.0))
            self.gan_patch = data["spatial_size"] // 4 + 2   # PartDiscriminator opens with a valid 3x3 conv (:87)
            self._gan_rng = torch.Generator().manual_seed(config["general"].get("seed", 42) + 7919)
            self._gan_off = torch.zeros(2, dtype=torch.int32, device=self.device)   # this step's window corner (oy, ox)
        # ---- data parallel (replaces nn.DataParallel, :213-214)
        self._pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if self.world > 1:
            # every rank seeds torch alike (same initial weights), but the ranks are different SAMPLES of the global batch:
            # their dropout masks and posterior noise (both hashed from this base) must differ
            ops.set_dropout_seed((config["general"].get("seed", 42) + 0x632BE5AB * dist.get_rank(process_group)) & 0xFFFFFFFF)
        broadcast_parameters(self.optimizer.buckets, 0, process_group)
        # backward finishes dd first, then ed/du, then eu: launch in that order
        self.averager = BucketedGradAverager(self.optimizer.buckets, process_group)
        # ---- schedules (:296-347)
        self.total_steps = total_steps if total_steps is not None else tr["end_iteration"]
        imax = tr["information_max"]
        if tr.get("imax_scaling", "none") == "ascend":
            self._imax = (0, imax)
        elif tr.get("imax_scaling") == "descend":
            self._imax = (imax, 0)
        else:
            self._imax = (imax, imax)
        self.lr = self._adjust_lr(0)
        self.imax = self._adjust_imax(0)
        for pg in self.optimizer.param_groups:
            pg["lr"] = self.lr
        self.gamma = torch.zeros((), device=self.device, dtype=torch.float32)  # device-resident controller state
        self._one = torch.ones((), device=self.device, dtype=torch.float32)    # seed of every backward pass
        # ---- hipGraph replay of the whole step (opt-in: ``training.hip_graph`` / VUNET_HIP_GRAPH=1 / the keyword)
        self._dev_sched = False
        self._graphs = {}
        # VGG target pass beside the small-map part of the generator's forward (tools/ab_step.py target_overlap)
        # (on wherever the trainer runs on several streams; measured -1.4 % .. -2.5 % step time)
        self._target_overlap = os.environ.get("VUNET_TARGET_OVERLAP", "1") != "0"
        self._target_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._capture_forks = os.environ.get("VUNET_CAPTURE_FORKS", "1") != "0"   # (A/B switch: the capture keeps the fork)
        # weight folds beside the VGG target pass: measured -0.4 % (tools/ab_step.py pack_overlap), inside box-to-box noise,
        # at the price of two concurrent streams in the part of the step the per-kernel roofline is read from: off
        self._pack_overlap = os.environ.get("VUNET_PACK_OVERLAP", "0") == "1"
        if hip_graph is None:
            hip_graph = bool(tr.get("hip_graph", os.environ.get("VUNET_HIP_GRAPH", "0") == "1"))
        if hip_graph:
            self.enable_hip_graph()
        print(f"Number of trainable params is {n_parameters(self.vunet)}")

    # ---- schedules
    def _adjust_lr(self, it):
        lr0 = self.config["training"]["lr"]
        return float(linear_var(it, 0, self.total_steps, lr0, 0, 0, lr0))

    def _adjust_imax(self, it):
        a, b = self._imax
        return float(linear_var(it, 0, self.total_steps, a, b, min(a, b), max(a, b)))

    def adjust_params(self, it):
        """:500-512: lr / imax schedules after every iteration; gamma rides in the param groups."""
        self.lr = self._adjust_lr(it)
        self.imax = self._adjust_imax(it)
        for pg in self.optimizer.param_groups:
            pg["lr"] = self.lr
            pg["gamma"] = self.gamma

    # ---- hipGraph mode ------------------------------------------------------------------------------------
    # A training step is ~1900 kernel launches issued from Python (~18 ms of host time); for the small
    # configurations (batch 1-2, 128x128 maps) the GPU finishes sooner than the host can issue.  In graph mode
    # everything that changes from step to step lives in DEVICE memory -- the learning rate and Adam's step count
    # (vunet_adam_step_dev), the dropout step counter (vunet_set_dropout_step), gamma and information_max of the KL
    # controller -- so the launch arguments of a step are constants, the step is captured once (after the
    # initialisation batches, per batch geometry) and replayed with one hipGraphLaunch.  The host still runs the
    # schedules (:500-512) and writes the two scalars before every replay.  Same arithmetic as the eager device-schedule
    # step, kernel for kernel (tests/test_hip_training.py::test_hip_graph_replay_is_bit_identical_to_eager).
    def enable_hip_graph(self, capture: bool = True):
        """``capture=False`` keeps the device-resident schedule but launches eagerly (the parity baseline of the test)."""
        if self.device.type != "cuda":
            raise RuntimeError("hipGraph mode needs the GPU")
        if self.gan is not None and self.gan.use_gp:
            raise RuntimeError("hipGraph mode does not cover the R1 penalty (a double backward assembled from ATen ops per "
                               "step); run that configuration eagerly")
        if self.averager.active and not self.averager.native:
            raise RuntimeError("hipGraph mode with data parallelism needs the gradient all-reduce on the C-ABI RCCL "
                               "communicator (vunet_dp_allreduce_bucket: an ordinary stream operation, capturable); this "
                               "averager runs on torch.distributed collectives -- run eagerly")
        if not self._dev_sched:
            self._lr_dev = torch.full((1,), self.lr, dtype=torch.float64, device=self.device)
            self._imax_dev = torch.full((), self.imax, dtype=torch.float32, device=self.device)
            self._drop_step = torch.zeros(1, dtype=torch.int32, device=self.device)
            self.optimizer.use_device_schedule(self._lr_dev)
            if self.train_regressor:
                self.optimizer_regressor.use_device_schedule()
            if self.gan is not None:
                self.gan.opt.use_device_schedule()
            ops.set_dropout_step(self._drop_step)
            self._graph_stream = torch.cuda.Stream()
            self._dev_sched = True
            self._eager_dev_steps = 0
        self._capture = bool(capture)
        return self

    def _graph_key(self, batch, it):
        tr = self.config["training"]
        # the KL term joins the loss (and the l2 layers stop initialising) after the init batches; two eager steps of
        # the final shape in THIS process come first, so that lazily built state (frozen VGG packs, the prepack table,
        # companion streams) exists before anything is recorded
        if it <= tr["n_init_batches"] or self._eager_dev_steps < 2:
            return None
        return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in batch.items() if torch.is_tensor(v)))

    def _train_fn_graph(self, batch, it, eps, reg_eps):
        # Every device-schedule step -- the eager ones before the capture, the capture itself and the replays -- runs on
        # ONE dedicated stream: the per-stream state the step builds lazily (weight-gradient companion streams, slab arenas
        # and the item tables of the batched weight-norm backward, which hold arena addresses and are uploaded from host
        # memory on first use) then already exists, with the same addresses, when the step is recorded.
        caller = torch.cuda.current_stream()
        gs = self._graph_stream
        gs.wait_stream(caller)
        with torch.cuda.stream(gs):
            out = self._train_fn_graph_on_stream(batch, it, eps, reg_eps)
        caller.wait_stream(gs)
        return out

    def _packed_models(self):
        """The models whose weight-norm folds are done once per step (ops.prepacked): the generator and, with the adversarial
        term, the discriminator -- it runs three times per step (for the generator's loss, on the real and on the fake patch)
        and re-folded its 8 layers each time (29 x 4 small launches, ~0.7 ms of the step's kernel time)."""
        return (self.vunet, self.gan.disc) if self.gan is not None else (self.vunet,)

    def _capture_agreed(self, ok_here: bool) -> bool:
        """True iff EVERY rank of the trainer's process group recorded the step (single process: this one did).  Reached when a
        rank meets a new graph key: batch shapes -- hence keys -- must be the same on every rank at every step (the
        data-parallel contract: equal per-rank batches), or the ranks would not enter this reduction together."""
        if not (self.averager.active and dist.is_initialized() and self.world > 1):
            return ok_here
        pg = getattr(self, "_pg", None)
        flag = torch.tensor([1 if ok_here else 0], device=self.device if dist.get_backend(pg) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
        return int(flag.item()) == 1

    def _train_fn_graph_on_stream(self, batch, it, eps, reg_eps):
        # host-side schedule values of THIS step -> device (one tiny launch outside the graph)
        ops.set_schedule_(self._lr_dev, self.lr, self._imax_dev, self.imax, self._drop_step, it & 0x7FFFFFFF)
        # the library keeps ONE process-wide counter pointer: another graph-mode trainer (or set_dropout_step(None)) may
        # have replaced it since this trainer's last step -- re-assert ours (host-only, no launch) before anything draws
        ops.set_dropout_step(self._drop_step)
        ops.set_dropout_host_step(0)            # (the device counter carries the step)
        ops.reset_dropout_counter()
        key = self._graph_key(batch, it) if (self._capture and eps is None and reg_eps is None) else None
        if key is None:
            if it > self.config["training"]["n_init_batches"]:
                self._eager_dev_steps += 1
            self.optimizer.zero_grad()
            with ops.prepacked(*self._packed_models()):
                return self._step(batch, it, eps, reg_eps)
        rec = self._graphs.get(key)
        fresh = rec is None
        if fresh:
            static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            # every optimiser whose step() runs inside the step keeps a host-side count the aborted pass would leave advanced
            opts = [self.optimizer] + ([self.optimizer_regressor] if self.train_regressor else []) + \
                   ([self.gan.opt] if self.gan is not None else [])
            steps_before = [[b.step for b in o.buckets] for o in opts]
            err = out = None
            try:
                with torch.cuda.graph(graph, stream=self._graph_stream):
                    self.optimizer.zero_grad()
                    with ops.prepacked(*self._packed_models()):
                        out = self._step(static, it, None, None)
            except Exception as e:   # noqa: BLE001 -- a runtime that cannot record this step: say so and issue eagerly from here on
                err = e
            # Data parallel: the ranks must stay in the same mode -- a rank that replays its recorded all-reduces while
            # another issues them eagerly after an aborted pass would pair collectives that do not belong together (or
            # wait for ones that never come).  One MIN over "my capture succeeded", outside the capture.
            if not self._capture_agreed(err is None):
                import sys
                why = f"{type(err).__name__}: {err}" if err is not None else "another rank could not record it"
                print(f"[vunet] hipGraph capture of the training step failed ({why}); the step is issued "
                      "eagerly (device-resident schedule) from here on", file=sys.stderr)
                self._capture = False
                self._graphs.clear()
                del graph
                for o, ns in zip(opts, steps_before):   # the aborted / discarded pass advanced the host-side counts
                    for b, n in zip(o.buckets, ns):
                        b.step = n
                ops.flush_weight_grads()
                torch.cuda.synchronize()
                # the recording pass ran the averager's hooks: its per-step state says "every bucket launched" -- start over,
                # or the eager pass would issue no all-reduce at all and finish() would refuse the step
                self.averager.start_step()
                ops.reset_dropout_counter()   # the recording pass drew this step's seeds: the eager pass draws the same ones
                self.optimizer.zero_grad()
                with ops.prepacked(*self._packed_models()):
                    return self._step(batch, it, eps, reg_eps)
            rec = self._graphs[key] = {"graph": graph, "static": static, "out": out}
        else:
            for k, v in batch.items():
                if torch.is_tensor(v):
                    rec["static"][k].copy_(v, non_blocking=True)
            # the capture pass already advanced the host-side step counts once
            self.optimizer.note_replayed_steps(1)
            if self.train_regressor and "reg_imgs" in batch:
                self.optimizer_regressor.note_replayed_steps(batch["reg_imgs"].shape[1])
            if self.gan is not None:
                self.gan.opt.note_replayed_steps(1)
        rec["graph"].replay()
        return dict(rec["out"])   # tensors are the graph's static outputs: valid until the next train_fn call

    # ---- one training step (:360-466)
    def train_fn(self, batch: Dict[str, torch.Tensor], eps=None, reg_eps=None) -> Dict[str, torch.Tensor]:
        """``eps`` / ``reg_eps`` inject the Gaussian draws of the posterior sampling (one tensor per latent scale; for
        the regressor side loop one such list per regressor step) -- the parity tests' hook, None in production."""
        if not self.vunet.training:   # (Module.train() walks all ~400 sub-modules: 1.5 ms per step)
            self.vunet.train()
        self.iteration += 1
        it = self.iteration
        self.averager.start_step()
        if self.gan is not None:
            # one (patch x patch) window per step, the same for the real and the generated batch: drawn on the host, handed
            # to the step through device memory (ops.CropWindow), so that the step's launch arguments never change
            P, S = self.gan_patch, batch["pose_img"].shape[-1]
            off = torch.randint(0, S - P + 1, (2,), generator=self._gan_rng).to(torch.int32)
            self._gan_off.copy_(off, non_blocking=True)
        with ops.kernel_noise():   # posterior noise from the sampling kernels themselves (ops.kernel_noise)
            return self._train_fn(batch, it, eps, reg_eps)

    def _train_fn(self, batch, it, eps, reg_eps):
        if self._dev_sched:
            out = self._train_fn_graph(batch, it, eps, reg_eps)
        else:
            if ops.dropout_step_counter() is not None:   # left behind by a graph-mode trainer of this process
                ops.set_dropout_step(None)
            # the seed sequence restarts every step and carries the step number as a host-side offset: the same dropout
            # masks and posterior noise as the device-schedule / captured step of this number draws
            ops.reset_dropout_counter()
            ops.set_dropout_host_step(it & 0x7FFFFFFF)
            self.optimizer.zero_grad()
            target_features = None
            side = self.vunet._side_stream
            if side is not None and self.device.type == "cuda" and self._pack_overlap:
                # the step's weight folds (two launches over every layer, ~0.3 ms with nothing else to run) go to the side
                # stream; the main stream meanwhile runs the one part of the step that needs none of them -- the frozen
                # VGG19's pass over the target image (lib/losses.py:88-92: no graph, no generator weights)
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    packed = ops.prepacked(*self._packed_models())
                    packed.__enter__()
                try:
                    with torch.no_grad():
                        target_features = self.custom_vgg.features_for_loss(batch["pose_img"])
                    main.wait_stream(side)
                    out = self._step(batch, it, eps, reg_eps, target_features)
                finally:
                    packed.__exit__(None, None, None)
            else:
                with ops.prepacked(*self._packed_models()):  # all weight-norm folds of the step in two launches
                    out = self._step(batch, it, eps, reg_eps)
        self.adjust_params(it)
        out.update({"learning_rate": self.lr, "gamma": self.gamma, "imax": self.imax})
        return out

    def _step(self, batch, it, eps, reg_eps=None, target_features=None):
        tr = self.config["training"]
        target_img = batch["pose_img"]
        shape_img = batch["stickman"]
        pose_img = batch.get("pose_img_inplane", target_img)
        # (inside a hipGraph capture too: ``wait_stream`` on a capturing stream forks the target stream into the capture, the
        # wait on it below joins it back -- the recorded graph keeps the branch)
        tstream = self._target_stream if (target_features is None and self._target_overlap and target_img.is_cuda
                                            and self.vunet._side_stream is not None
                                            and (self._capture_forks or not torch.cuda.is_current_stream_capturing())) else None
        if tstream is not None:
            # The frozen VGG19's pass over the TARGET image needs nothing from the generator.  It is issued on a stream of
            # its own once the appearance pyramid is through, so that its chip-filling kernels run beside the bottleneck
            # and the decoder's 4x4 .. 32x32 levels -- ~50 launches of a few microseconds each that leave the GPU idle.
            main = torch.cuda.current_stream()
            box = {}

            def start_target_pass():
                tstream.wait_stream(main)
                with torch.cuda.stream(tstream), torch.no_grad():
                    box["f"] = self.custom_vgg.features_for_loss(target_img)
            out_img, means, logstds, _ = self.vunet(pose_img, shape_img, eps, after_encoder=start_target_pass)
            main.wait_stream(tstream)
            target_features = box["f"]
            for t_ in target_features.values():
                t_.record_stream(main)
        else:
            out_img, means, logstds, _ = self.vunet(pose_img, shape_img, eps)
        ld = vgg_loss(self.custom_vgg, target_img, out_img, target_features=target_features)
        kl = compute_kl_with_prior(means, logstds)
        # :391-405 -- likelihood_loss = ll_weight * sum(terms); loss = likelihood_loss (+ tuning * kl after the init batches)
        tuning = 1.0 if self.config["architecture"].get("cvae", False) else self.gamma
        loss, likelihood_loss = ops.TotalLoss.apply(kl, tuning, tr["ll_weight"], it > tr["n_init_batches"],
                                                    *[ld[k] for k in ld])
        out = {}
        if self.train_regressor and "reg_imgs" in batch:
            loss_regressor = self._regressor_steps(batch, reg_eps)
            # pure scalar offset: no gradient path to the VUnet (:413 runs under no_grad)
            loss = loss - torch.clamp(loss_regressor.detach(), max=1.2) * tr["weight_regressor"]
            out["loss_reg"] = loss_regressor.detach()
        patches = None
        if self.gan is not None:
            P = self.gan_patch
            fake_patch = ops.CropWindow.apply(out_img, self._gan_off, P)
            real_patch = ops.CropWindow.apply(target_img, self._gan_off, P)
            gen_loss, w = self.gan.get_genloss(fake_patch, likelihood_loss, self.vunet.dd.out_conv.conv.weight_v)
            loss = loss + self.gan_weight * w * gen_loss
            out["gen_loss"] = gen_loss.detach()
            patches = (real_patch.detach(), fake_patch.detach())
        loss.backward(gradient=self._one)   # (a resident 1.0: no ones_like launch per step)
        self.averager.mark_backward_end()
        ops.flush_weight_grads()   # (normally already done by the end-of-backward callback) before the streams are joined
        self.vunet.join_streams()
        ops.join_wgrad_streams()
        kl_avg = self.averager.finish(kl.detach().clone().reshape(1)) if self.averager.active else kl.detach()
        self.optimizer.step()
        if patches is not None:
            out.update(self.gan.train_disc(*patches, as_tensors=True))   # device scalars: no host synchronisation
        # gamma controller on the device (:82-85,442); with DP every rank sees the averaged KL
        if self._dev_sched:   # in place, information_max from device memory: nothing here is a per-step launch argument
            ops.gamma_update_(self.gamma, self._imax_dev, kl_avg, tr["gamma_step"])
        else:
            self.gamma = torch.clamp(self.gamma - tr["gamma_step"] * (self.imax - kl_avg.reshape(())), min=0.0)
        # The scalars that LEAVE the step are gathered into one small tensor of their own (one launch): the loss kernels'
        # outputs are 4-byte views into the step's 512 KB accumulator arena (ops._zero_scalar), and a caller that keeps the
        # returned values -- a logging list -- would otherwise keep one arena alive per step.
        names = ["loss", "likelihood_loss", "kl_loss"] + list(ld)
        terms = [loss, likelihood_loss, kl] + [ld[k] for k in ld]
        packed = torch.cat([t.detach().reshape(1) for t in terms])
        out.update({k: packed[i].view(t.shape) for i, (k, t) in enumerate(zip(names, terms))})
        return out

    def _regressor_steps(self, batch, reg_eps=None):
        """:407-425.  The reference runs ``ed(eu(reg_imgs[:, i]))`` inside the loop; the encoder is frozen there
        (``no_grad``, no VUnet update until the outer step), so the ``reg_steps`` encoder passes do not depend on the
        regressor updates between them and run here as ONE batch of ``B * reg_steps`` samples -- the same per-sample
        computation, far fewer (and fuller) launches on the small maps.  The regressor steps stay sequential."""
        reg_imgs, reg_targets = batch["reg_imgs"], batch["reg_targets"]
        b, steps = reg_imgs.shape[:2]
        with torch.no_grad():
            flat = reg_imgs.transpose(0, 1).reshape(steps * b, *reg_imgs.shape[2:]).contiguous()   # step-major
            eps_all = None if reg_eps is None else [torch.cat([e[s_] for e in reg_eps], dim=0)
                                                     for s_ in range(len(reg_eps[0]))]
            _, means_all, _, _ = self.vunet.ed(self.vunet.eu(flat), eps_all)
        loss_regressor = None
        for i in range(steps):
            means = [m[i * b:(i + 1) * b] for m in means_all]
            # (the regressor's four plain layers -- two full-window embedders run as 1x1 products of the flattened latent, two
            # linears -- pack their weights per call: ~0.5 ms of small launches per step that ops.prepacked does not cover, the
            # embedders' weight view has no owner module whose layout the batched pack knows)
            preds = self.regressor(means)
            tgts = reg_targets[:, i].reshape(reg_targets.shape[0], -1)
            loss_regressor = torch.norm(preds - tgts, dim=1).mean()
            self.optimizer_regressor.zero_grad()
            loss_regressor.backward()
            self.optimizer_regressor.step()
        return loss_regressor

    @torch.no_grad()
    def transfer(self, app_img, stickman, dtype: str = "f32"):
        """Inference path used by the render loop (models/vunets.py:508-515); ``dtype="bf16"``: BASELINE config 5."""
        self.vunet.eval()
        with ops.inference_precision(dtype):
            return self.vunet.transfer(app_img, stickman)

    # ---- checkpoint layout of :471-494: {"model", "optimizer"} (the "reg_ckpt" file); the regressor and its optimiser
    # form the reference's second file ("regressor"), here the "regressor" entry of the same dict
    def state_dict(self):
        opt = self.optimizer.state_dict()
        gamma = float(self.gamma)   # the reference keeps gamma as a host number in every param group (:507-512)
        for g in opt["param_groups"]:
            g["gamma"] = gamma
        sd = {"model": self.vunet.state_dict(), "optimizer": opt}
        if self.train_regressor:
            sd["regressor"] = {"model": self.regressor.state_dict(), "optimizer": self.optimizer_regressor.state_dict()}
        if self.gan is not None:   # DiscTrainer's own save dict (models/synth_discriminator.py:228-231) rides along
            sd["discriminator"] = self.gan.checkpoint()
        return sd

    def load_state_dict(self, ckpt):
        """Restart as :87-95, :248-255 do: weights, Adam state, the iteration from Adam's step count, gamma from the
        optimiser's param groups, then the lr / imax schedules re-derived for that iteration."""
        self.vunet.load_state_dict(ckpt["model"])
        if "optimizer" in ckpt and ckpt["optimizer"] is not None:
            op = ckpt["optimizer"]
            gamma = None
            for g in op["param_groups"]:
                if "gamma" in g:
                    gamma = g["gamma"]
            self.optimizer.load_state_dict(op)
            states = list(op["state"].values())
            if states:
                self.iteration = int(states[-1]["step"])  # :248-255
            if gamma is not None:
                self.gamma.fill_(float(gamma))   # in place: a captured step holds this tensor's address
            self.adjust_params(self.iteration)
        if self.train_regressor and ckpt.get("regressor") is not None:
            self.regressor.load_state_dict(ckpt["regressor"]["model"])
            if ckpt["regressor"].get("optimizer") is not None:
                self.optimizer_regressor.load_state_dict(ckpt["regressor"]["optimizer"])
        if self.gan is not None and "discriminator" in ckpt:
            self.gan.disc.load_state_dict(ckpt["discriminator"]["disc"])
            self.gan.opt.load_state_dict(ckpt["discriminator"]["opt"])


def synthetic_keypoints(batch_size: int, spatial_size: int, n_keypoints: int = 17, seed: int = 42, rank: int = 0):
    """Synthetic 17-joint skeletons of SURVEY 8(d): joints ~ N(frame centre, (40 px * size / 256)^2), clipped to the
    frame.  CPU tensor [B, J, 2] (x, y) in pixels."""
    g = torch.Generator().manual_seed(seed + 1000 * rank + 17)
    c, sd = 0.5 * spatial_size, 40.0 * spatial_size / 256.0
    kps = torch.randn(batch_size, n_keypoints, 2, generator=g) * sd + c
    return kps.clamp_(0.0, float(spatial_size - 1))


def synthetic_batch(batch_size: int, spatial_size: int, device, seed: int = 42, n_channels_x: int = 3,
                    with_regressor: bool = False, reg_steps: int = 5, n_keypoints: int = 17, rank: int = 0,
                    stickman: str = "raster", appearance_size: Optional[int] = None):
    """Synthetic pose + appearance batch of SURVEY 8(d): U(-1,1) target image; the stickman is drawn by the GPU
    rasteriser (csrc/raster.hip, lib/utils.make_joint_img_batch) from synthetic_keypoints -- planes with the levels
    {0, 127, 255} / 255 * 2 - 1 like the dataset's (data/base_dataset.py:183-190) -- or, with ``stickman="mask"`` (CPU
    tensors, no GPU at hand), a sparse {-1, +1} mask with 5 % ones.  ``n_channels_x != 3``: the multi-part appearance
    input of the DeepFashion / Market configs, ``appearance_size`` square (default: half the image, box_factor 1)."""
    g = torch.Generator().manual_seed(seed + 1000 * rank)
    pose = torch.rand(batch_size, 3, spatial_size, spatial_size, generator=g) * 2 - 1
    mask = (torch.rand(batch_size, 3, spatial_size, spatial_size, generator=g) < 0.05).float() * 2 - 1
    if stickman == "raster" and torch.device(device).type == "cuda":
        from ..lib.utils import make_joint_img_batch
        kps = synthetic_keypoints(batch_size, spatial_size, n_keypoints, seed, rank).to(device)
        stick = make_joint_img_batch((spatial_size, spatial_size), kps)
    else:
        stick = mask.to(device)
    batch = {"pose_img": pose.to(device), "stickman": stick}
    if n_channels_x != 3:
        xs = appearance_size or spatial_size // 2
        batch["pose_img_inplane"] = (torch.rand(batch_size, n_channels_x, xs, xs, generator=g) * 2 - 1).to(device)
    if with_regressor:
        batch["reg_imgs"] = (torch.rand(batch_size, reg_steps, 3, spatial_size, spatial_size, generator=g) * 2 - 1
                             ).to(device)
        batch["reg_targets"] = torch.rand(batch_size, reg_steps, n_keypoints, 2, generator=g).to(device)
    return batch
