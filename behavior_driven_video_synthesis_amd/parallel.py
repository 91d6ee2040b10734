"""Pure data parallelism for the VUnet training step: one process per GPU, RCCL over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (experiments/shape_and_pose_net.py:213-214,
223-224, 230-233), whose per-step parameter broadcast, input scatter and output gather disappear:
each rank keeps a persistent replica and its own shard of the batch, and the only exchange is the
gradient average -- an all-reduce per flat parameter bucket (optim.FlatBucket: dd / ed / du / eu),
launched asynchronously as soon as the last gradient of the bucket has been written, so it overlaps
with the rest of the backward pass.  ``backend="nccl"`` is RCCL on ROCm; the same code runs on gloo
for the CPU tests.

On the GPU the bucket all-reduces of a step go through the C-ABI library's own RCCL communicator
(``vunet_dp_init`` / ``vunet_dp_allreduce_bucket``, csrc/dp_rccl.hip): torch.distributed remains the
bootstrap -- rendezvous, the one-time parameter broadcast, the 128-byte unique id, the end-of-run consistency
check -- while the data path of every step is an ordinary operation on the communication stream, which also makes
it capturable in a hipGraph.  ``VUNET_DP_BACKEND=torch`` keeps the step's collectives on torch.distributed; if
the native communicator cannot be brought up the averager says so on stderr and does the same.
"""
from __future__ import annotations

import collections
import os
from typing import List, Optional

import torch
import torch.distributed as dist


class BucketedGradAverager:
    def __init__(self, buckets, process_group=None, overlap: bool = True):
        self.buckets = list(buckets)
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # VUNET_DP_FORCE=1: run the full hook / async all-reduce machinery even with one rank (1-GPU smoke test)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("VUNET_DP_FORCE") == "1")
        self.overlap = overlap
        self._works = []
        self._comm_stream = None          # HIP stream the bucket all-reduces are ordered on (GPU only)
        self._events = collections.deque(maxlen=16)   # per step: [(start, end) HIP events on that stream, one pair per collective]
        self._bwd_end = collections.deque(maxlen=16)  # per step: the HIP event recorded where backward ended (or None)
        self._pending: List[int] = []
        self._expected: List[Optional[int]] = [None] * len(self.buckets)  # learnt on the first step
        self._fired: List[int] = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._fired_at_launch = [0] * len(self.buckets)
        # Collectives pair up across ranks by ISSUE ORDER, so every rank must launch the bucket all-reduces in the same
        # order whatever order its own backward completes them in: the order is learnt on the first step (completion
        # order on rank 0, broadcast once) and from then on a bucket is launched only after its predecessors in it.
        self._order: Optional[List[int]] = None
        self._next_pos = 0
        self._ready = [False] * len(self.buckets)
        self._last_fire_seq = [1 << 60] * len(self.buckets)
        self._seq = 0
        self.native = False               # bucket all-reduces through the C-ABI RCCL communicator (GPU only)
        self.backend = "none"
        if self.active:
            self.backend = "torch.distributed"
            if self.buckets[0].grad.is_cuda and os.environ.get("VUNET_DP_BACKEND", "native") != "torch":
                self.native = _init_native_comm(process_group)
                if self.native:
                    self.backend = "rccl-cabi"
        if self.active:
            from . import ops
            self._bucket_of = {}
            for bi, b in enumerate(self.buckets):
                for p in b.params:
                    self._bucket_of[id(p)] = bi
                    # gradients that autograd accumulates itself ...
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))
            # ... and those the HIP weight-norm backward writes straight into the flat bucket
            ops.add_grad_hook(self._direct_hook)

    def _make_hook(self, bi):
        def hook(param):
            self._fired[bi] += 1
            self._seq += 1
            self._last_fire_seq[bi] = self._seq
            if self.overlap and self._expected[bi] is not None and self._fired[bi] == self._expected[bi]:
                self._ready[bi] = True
                self._launch_ready()
        return hook

    def _direct_hook(self, param):
        bi = self._bucket_of.get(id(param))
        if bi is not None:
            self._make_hook(bi)(param)

    def _launch_ready(self):
        """Launch, in the agreed order, every bucket that is complete and whose predecessors have been launched."""
        if self._order is None:
            return
        while self._next_pos < len(self._order) and self._ready[self._order[self._next_pos]]:
            self._launch(self._order[self._next_pos])
            self._next_pos += 1

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        self._fired_at_launch[bi] = self._fired[bi]
        b = self.buckets[bi]
        b.gather_foreign_grads()
        self._all_reduce(b.grad)

    def _all_reduce(self, t: torch.Tensor):
        """Asynchronous SUM all-reduce of ``t``.  On the GPU the collective is ordered on ONE dedicated communication
        stream, whichever stream the calling autograd hook happens to run on: that stream first waits for the caller's
        stream and for the weight-gradient companion streams (the in-place gradient writes), RCCL's own stream is chained
        to it by ``Work.wait()``, and ``finish()`` makes the optimiser's stream wait for it.  A pair of HIP events on the
        communication stream brackets every collective (``mean_allreduce_ms``)."""
        if not t.is_cuda:
            self._works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
            return
        from . import ops
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        ops.join_wgrad_streams()   # cur now waits for the companion streams' in-place gradient writes
        self._comm_stream.wait_stream(cur)
        timed = not torch.cuda.is_current_stream_capturing()   # (timing events cannot be read back from a captured graph)
        with torch.cuda.stream(self._comm_stream):
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if self.native:
                t.record_stream(self._comm_stream)
                if os.environ.get("VUNET_DP_DRYRUN") != "1":   # (debug: the stream topology without the collective)
                    ops._call("vunet_dp_allreduce_bucket", ops._p(t), t.numel(), 0, ops._stream())
            else:
                work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                work.wait()        # stream-level: the communication stream waits for RCCL's stream, the host does not block
            if timed:
                e1.record()
        if timed and self._events:
            self._events[-1].append((e0, e1))

    def start_step(self):
        self._fired = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._works = []
        # (a step replayed from a captured hipGraph issues nothing from Python and records no events: its empty record is
        #  reused, so the measurements of the eagerly issued steps before the capture stay in the window)
        if not self._events or self._events[-1]:
            self._events.append([])
            self._bwd_end.append(None)
        self._next_pos = 0
        self._ready = [False] * len(self.buckets)
        self._seq = 0
        self._last_fire_seq = [1 << 60] * len(self.buckets)

    def mark_backward_end(self):
        """Called by the training step right after ``loss.backward()``: a HIP event on the current stream that marks where
        backward ended, against which ``overlap_fraction`` places the collectives."""
        if (self.active and self._bwd_end and torch.cuda.is_available() and self.buckets[0].grad.is_cuda
                and not torch.cuda.is_current_stream_capturing()):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._bwd_end[-1] = ev

    def finish(self, extra_scalars: Optional[torch.Tensor] = None):
        """Flush buckets that were not launched from the hooks, wait for all, scale by 1/world.

        ``extra_scalars`` (e.g. the KL value feeding the gamma controller) is averaged too so every
        rank runs the identical controller (experiments/shape_and_pose_net.py:82-85,442).
        """
        if not self.active:
            return extra_scalars
        for bi in range(len(self.buckets)):
            if self._launched[bi] and self._fired[bi] != self._fired_at_launch[bi]:
                # a gradient landed after the bucket's all-reduce had been launched from the hooks: the firing pattern
                # of backward changed between steps and the reduced values are incomplete -- never continue silently
                raise RuntimeError(f"bucket {self.buckets[bi].name!r}: {self._fired[bi]} gradient writes this step, its "
                                   f"all-reduce was launched after {self._fired_at_launch[bi]} (pattern learnt earlier); "
                                   "construct the averager with overlap=False for graphs that change between steps")
            self._expected[bi] = self._fired[bi]   # learnt on the first step, re-learnt if backward fired fewer hooks
        if self._order is None:
            # first step: agree on ONE launch order -- rank 0's completion order (buckets without gradients last)
            order = sorted(range(len(self.buckets)), key=lambda i: (self._last_fire_seq[i], i))
            t = torch.tensor(order, dtype=torch.int64, device=self.buckets[0].grad.device)
            if self.world > 1:
                dist.broadcast(t, src=0, group=self.pg)
            self._order = [int(v) for v in t.tolist()]
        for bi in self._order:                     # everything the hooks have not launched, in the agreed order
            self._launch(bi)
        if extra_scalars is not None:
            self._all_reduce(extra_scalars)
        for w in self._works:
            w.wait()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        inv = 1.0 / self.world
        for b in self.buckets:
            b.grad.mul_(inv) if not b.grad.is_cuda else _scale_(b.grad, inv)
        if extra_scalars is not None:
            extra_scalars.mul_(inv)
        return extra_scalars


    def mean_allreduce_ms(self):
        """Mean per-step time the communication stream spent in the gradient all-reduces (HIP events around every
        collective, summed per step); None without a process group.  Synchronises the device."""
        steps = [ev for ev in self._events if ev]
        if not self.active or not steps:
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for ev in steps for a, b in ev) / len(steps)


    def overlap_fraction(self):
        """Fraction of the gradient all-reduce time that ran BEFORE backward had finished on the compute stream, over
        the recorded steps (HIP events: each collective's start / end on the communication stream against the
        end-of-backward marker).  None without a process group or without markers.  Synchronises the device."""
        if not self.active:
            return None
        torch.cuda.synchronize()
        tot = hidden = 0.0
        for evs, end in zip(self._events, self._bwd_end):
            if end is None:
                continue
            for e0, e1 in evs:
                dur = e0.elapsed_time(e1)
                tot += dur
                hidden += min(max(e0.elapsed_time(end), 0.0), dur)
        return hidden / tot if tot > 0 else None

    def replica_checksums(self):
        """Bit-exact checksum of every flat parameter bucket (the int32 view summed in int64): equal across ranks iff the
        replicas hold identical parameters."""
        return torch.stack([b.flat.view(torch.int32).to(torch.int64).sum() for b in self.buckets])

    def replicas_consistent(self) -> Optional[bool]:
        """All-gather the bucket checksums and compare: True iff every rank holds bit-identical parameters -- the
        invariant of synchronous data parallelism (same initial broadcast, same averaged gradients, same Adam)."""
        if not dist.is_initialized():
            return None
        mine = self.replica_checksums()
        world = dist.get_world_size(self.pg)
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=self.pg)
        return all(bool(torch.equal(g, gathered[0])) for g in gathered)


_native_comm = {"world": 0}


def _init_native_comm(process_group=None) -> bool:
    """Bring up the C-ABI library's RCCL communicator over the ranks of ``process_group`` (once per process; a second
    averager of the same world size -- the discriminator's -- shares it).  The unique id travels through torch.distributed."""
    from . import _lib
    lib = _lib.lib()
    world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
    if _native_comm["world"] == world and lib.vunet_dp_world() == world:
        return True
    if lib.vunet_dp_world() != 0:
        return False                      # a communicator of another shape exists already: stay on torch.distributed
    import ctypes
    import sys
    flag_dev = "cuda" if dist.get_backend(process_group) == "nccl" else "cpu"

    def all_agree(ok_here: bool) -> bool:
        ok = torch.tensor([1 if ok_here else 0], device=flag_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=process_group)
        return int(ok.item()) == 1

    # Handshake in two rounds.  ncclCommInitRank blocks until EVERY rank has entered it, so a rank that cannot even bind
    # RCCL (dlopen / dlsym fails there) must be found out before anyone calls it: first every rank says whether the
    # library binds (vunet_dp_available: no GPU call, no communicator) and the flags are MIN-reduced over torch.distributed;
    # only if all can does rank 0 draw the unique id and everyone joins.  The second MIN (after init) catches a failed join.
    can_bind = lib.vunet_dp_available() == 1
    if not all_agree(can_bind):
        print(f"[vunet dp] rank {rank}: RCCL cannot be bound on every rank (here: {'yes' if can_bind else 'no'}); the step's "
              "all-reduces stay on torch.distributed", file=sys.stderr)
        return False
    buf = ctypes.create_string_buffer(128)
    rc = lib.vunet_dp_unique_id(buf) if rank == 0 else 0
    box = [bytes(buf.raw) if rc == 0 else None]
    src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
    dist.broadcast_object_list(box, src=src, group=process_group)
    if box[0] is None:          # rank 0 could not draw an id: every rank sees None and nobody enters the init
        print(f"[vunet dp] rank {rank}: no RCCL unique id from rank 0; the step's all-reduces stay on torch.distributed",
              file=sys.stderr)
        return False
    rc = lib.vunet_dp_init(world, rank, ctypes.c_char_p(box[0]))
    if not all_agree(rc == 0):      # all ranks or none
        if rc == 0:
            lib.vunet_dp_finalize()
        print(f"[vunet dp] rank {rank}: native RCCL communicator not available (code {rc}); the step's all-reduces stay on "
              "torch.distributed", file=sys.stderr)
        return False
    _native_comm["world"] = world
    return True


def shutdown_native_comm():
    """Destroy the C-ABI RCCL communicator (end of the run, before the process group goes away)."""
    if _native_comm["world"]:
        from . import _lib
        torch.cuda.synchronize()
        _lib.lib().vunet_dp_finalize()
        _native_comm["world"] = 0


def _scale_(t: torch.Tensor, a: float):
    from . import ops
    ops._call("vunet_axpby", ops._p(t), None, ops._p(t), float(a), 0.0, t.numel(), ops._stream())


def broadcast_parameters(buckets, src: int = 0, process_group=None):
    """One-time replica initialisation (replaces DataParallel's per-step broadcast)."""
    if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or os.environ.get("VUNET_DP_FORCE") == "1"):
        for b in buckets:
            dist.broadcast(b.flat, src=src, group=process_group)
