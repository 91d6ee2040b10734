"""CPU, gloo, world_size 2: the data-parallel gradient path (parallel.BucketedGradAverager over
optim.FlatBucket) -- the mean over ranks of the per-rank gradients equals the gradient of the
concatenated batch (every loss is a batch mean), bucket launches overlap with backward once
the firing pattern is known, dead parameters do not stall a bucket, scalars are averaged."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Bucket:
    """CPU stand-in with the FlatBucket interface (FlatBucket itself is device-agnostic; used directly too)."""


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from behavior_driven_video_synthesis_amd.optim import FlatBucket
    from behavior_driven_video_synthesis_amd.parallel import BucketedGradAverager, broadcast_parameters

    torch.manual_seed(100 + rank)  # different init per rank: broadcast must make the replicas identical
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    dead = torch.nn.Parameter(torch.ones(5))  # never receives a gradient (like ed.fin_block upstream)
    buckets = [FlatBucket(list(net[2].parameters()), "late"), FlatBucket(list(net[0].parameters()) + [dead], "early")]
    broadcast_parameters(buckets, 0)
    avg = BucketedGradAverager(buckets)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    per = 8 // world
    xs, ys = x_all[rank * per:(rank + 1) * per], y_all[rank * per:(rank + 1) * per]
    launched_early = []
    for step in range(3):
        avg.start_step()
        for b in buckets:
            b.zero_grad()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        launched_early.append(list(avg._launched))
        kl = avg.finish(loss.detach().clone().reshape(1))
    # reference: the whole batch on one replica with the broadcast weights
    ref = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    ref.load_state_dict(net.state_dict())
    full = ((ref(x_all) - y_all) ** 2).mean()
    full.backward()
    err = max(float((p.grad - r.grad).abs().max()) for p, r in zip(net.parameters(), ref.parameters()))
    # a step whose backward fires MORE gradient writes than the learnt pattern (two backward passes): the hooks launch
    # the all-reduce after the first pass, the second pass lands late -- finish() has to refuse, not average half a gradient
    avg.start_step()
    for b in buckets:
        b.zero_grad()
    (((net(xs) - ys) ** 2).mean() * 0.5).backward()
    (((net(xs) - ys) ** 2).mean() * 0.5).backward()
    try:
        avg.finish()
        refused = False
    except RuntimeError:
        refused = True
    for w in avg._works:
        w.wait()
    q.put((rank, err, float(kl), float(full), launched_early, float(dead.grad.abs().sum()),
           [float(p.detach().sum()) for p in net.parameters()], refused))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gradient_average_equals_full_batch_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, err, kl, full, launched, dead_grad, sums, refused in res:
        assert err < 1e-6, (rank, err)
        assert abs(kl - full) < 1e-6          # scalar averaged over ranks == full-batch mean loss
        assert dead_grad == 0.0
        assert launched[0] == [False, False]  # first step: firing pattern unknown, everything flushed in finish()
        assert launched[2][0] is True         # later steps: the bucket of the last layer is launched from backward
        assert refused                        # a changed firing pattern is an error, never a silently partial average
    assert res[0][6] == res[1][6]             # replicas identical after broadcast


def _worker_uneven(rank, world, port, q):
    """Four ranks whose backward passes complete the buckets in DIFFERENT orders (two independent sub-networks, two
    backward calls per step, the call order swapped on the odd ranks): collectives pair up by issue order, so the averager
    must launch them in one agreed order on every rank -- and the replicas must stay bit-identical."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from behavior_driven_video_synthesis_amd.optim import FlatBucket
    from behavior_driven_video_synthesis_amd.parallel import BucketedGradAverager, broadcast_parameters

    torch.manual_seed(7 + rank)
    net_a, net_b = torch.nn.Linear(5, 4), torch.nn.Linear(3, 7)   # buckets of different sizes: a mismatched pairing cannot pass
    buckets = [FlatBucket(list(net_a.parameters()), "a"), FlatBucket(list(net_b.parameters()), "b")]
    broadcast_parameters(buckets, 0)
    avg = BucketedGradAverager(buckets)
    g = torch.Generator().manual_seed(3)
    xa, xb = torch.randn(4 * world, 5, generator=g), torch.randn(4 * world, 3, generator=g)
    sl = slice(4 * rank, 4 * rank + 4)
    early = []
    for step in range(4):
        avg.start_step()
        for b in buckets:
            b.zero_grad()
        la, lb = net_a(xa[sl]).pow(2).mean(), net_b(xb[sl]).pow(2).mean()
        for loss in ((la, lb) if rank % 2 == 0 else (lb, la)):
            loss.backward()
        early.append(list(avg._launched))
        avg.finish()
        with torch.no_grad():
            for b in buckets:
                b.flat.add_(b.grad, alpha=-0.1)
    ref_a, ref_b = torch.nn.Linear(5, 4), torch.nn.Linear(3, 7)
    consistent = avg.replicas_consistent()
    # full-batch check of the last step's averaged gradient
    ref_a.load_state_dict(net_a.state_dict())
    ref_b.load_state_dict(net_b.state_dict())
    q.put((rank, consistent, avg._order, early, [float(b.flat.double().sum()) for b in buckets]))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_four_ranks_with_uneven_completion_order_launch_in_one_agreed_order():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    orders = {tuple(r[2]) for r in res}
    assert len(orders) == 1                                  # one launch order on every rank (rank 0's: bucket a first)
    assert orders.pop() == (0, 1)
    for rank, consistent, order, early, sums in res:
        assert consistent is True                            # bit-identical replicas after four optimiser steps
        assert early[0] == [False, False]                    # step 1: pattern unknown
        # later steps: the bucket that is first in the agreed order is launched from backward on the ranks that finish it
        # first; on the odd ranks (b finishes first) b has to WAIT for a, so both launch once a is complete
        assert early[3] == [True, True] if rank % 2 else early[3][0] is True
    assert len({tuple(r[4]) for r in res}) == 1


def _worker_capture_fallback(rank, world, port, q):
    """The protocol of a failed hipGraph capture under data parallelism (ShapePoseNet._train_fn_graph_on_stream), on gloo:
    (i) the ranks agree on "everybody recorded the step" with one MIN -- a rank whose own capture succeeded goes eager too
    when another's failed; (ii) the aborted recording pass leaves the averager saying "every bucket launched": re-running the
    step without ``start_step()`` issues no all-reduce and ``finish()`` refuses it; with it the step is a normal one and the
    replicas stay bit-identical (what bench.py's ``dp_consistent`` checks at the end of a run)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from behavior_driven_video_synthesis_amd.experiments.shape_and_pose_net import ShapePoseNet
    from behavior_driven_video_synthesis_amd.optim import FlatBucket
    from behavior_driven_video_synthesis_amd.parallel import BucketedGradAverager, broadcast_parameters

    torch.manual_seed(5 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3))
    buckets = [FlatBucket(list(net[2].parameters()), "late"), FlatBucket(list(net[0].parameters()), "early")]
    broadcast_parameters(buckets, 0)
    avg = BucketedGradAverager(buckets)
    fake = types.SimpleNamespace(averager=avg, world=world, device=torch.device("cpu"))
    agreed = [ShapePoseNet._capture_agreed(fake, True),            # everybody recorded it
              ShapePoseNet._capture_agreed(fake, rank != 1),       # rank 1 could not: nobody replays
              ShapePoseNet._capture_agreed(fake, False)]
    g = torch.Generator().manual_seed(11)
    x_all, y_all = torch.randn(8, 6, generator=g), torch.randn(8, 3, generator=g)
    per = 8 // world
    xs, ys = x_all[rank * per:(rank + 1) * per], y_all[rank * per:(rank + 1) * per]

    def step(restart):
        if restart:
            avg.start_step()
        for b in buckets:
            b.zero_grad()
        ((net(xs) - ys) ** 2).mean().backward()
        avg.finish()
        with torch.no_grad():
            for b in buckets:
                b.flat.add_(b.grad, alpha=-0.1)
    step(True)
    step(True)
    # the state an aborted recording pass leaves behind: hooks fired and every bucket "launched", nothing exchanged
    avg.start_step()
    avg._fired = list(avg._expected)
    avg._launched = [True] * len(buckets)
    avg._fired_at_launch = list(avg._expected)
    avg._next_pos = len(buckets)
    try:
        step(False)
        refused = False
    except RuntimeError:
        refused = True
    step(True)                      # what the trainer does: start the step over, then issue it eagerly
    step(True)
    q.put((rank, agreed, refused, avg.replicas_consistent(), [float(b.flat.double().sum()) for b in buckets]))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_failed_capture_protocol_keeps_the_ranks_in_one_mode_and_the_replicas_consistent():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_capture_fallback, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, agreed, refused, consistent, sums in res:
        assert agreed == [True, False, False], (rank, agreed)     # the same answer on both ranks, whoever failed
        assert refused                                            # without start_step() the re-run is refused, not half-averaged
        assert consistent is True
    assert res[0][4] == res[1][4]


def test_flat_bucket_views_and_adam_state_dict_layout():
    sys.path.insert(0, ROOT)
    from behavior_driven_video_synthesis_amd.optim import FlatBucket
    lin = torch.nn.Linear(4, 3)
    w0 = lin.weight.detach().clone()
    b = FlatBucket(list(lin.parameters()), "g")
    assert b.numel == 15 and torch.equal(lin.weight, w0)
    assert lin.weight.data_ptr() == b.flat.data_ptr() and lin.weight.grad.data_ptr() == b.grad.data_ptr()
    lin(torch.ones(2, 4)).sum().backward()
    assert float(b.grad.abs().sum()) > 0 and lin.weight.grad.data_ptr() == b.grad.data_ptr()
    b.zero_grad()
    assert float(b.grad.abs().sum()) == 0
