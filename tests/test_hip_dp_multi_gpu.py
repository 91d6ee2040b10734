"""-m gpu, needs >= 2 GPUs (skips cleanly on a 1-GPU box): the data-parallel training step over RCCL.

Two fresh child processes per GPU pair are started BEFORE this process touches a GPU (``device_count`` does not
initialise one): ``torchrun --nproc-per-node 2 tools/dp_check.py`` -- gradient buckets all-reduced from backward's
hooks on the communication stream, weight-gradient companion streams on -- and one single-rank run on the concatenated
batch.  Asserts: both ranks end with bit-identical parameters, gamma and losses histories are consistent, and the
replicas equal the single-rank result within fp32 tolerance (mean of per-rank gradients == full-batch gradient,
SURVEY 8e / F6; replaces nn.DataParallel of experiments/shape_and_pose_net.py:213-214)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rccl_run_matches_the_single_rank_run(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = str(tmp_path)
    script = os.path.join(ROOT, "tools", "dp_check.py")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", script, "--out", out],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([sys.executable, script, "--single", "--world", "2", "--out", out], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    r0, r1 = torch.load(os.path.join(out, "rank0.pt")), torch.load(os.path.join(out, "rank1.pt"))
    single = torch.load(os.path.join(out, "single.pt"))
    for a, b in zip(r0["flat"], r1["flat"]):
        assert torch.equal(a, b)                       # replicas stay bit-identical
    assert r0["gamma"] == r1["gamma"]                   # every rank ran the same gamma controller (averaged KL)
    assert r0["allreduce_ms"] is not None and r0["allreduce_ms"] > 0
    for a, s in zip(r0["flat"], single["flat"]):
        d = (a - s).abs().max().item()
        assert d <= 2e-4 * s.abs().max().item() + 1e-6, d
    # the rank-mean of the per-rank losses is the full-batch loss (every term is a batch mean)
    for la, lb, ls in zip(r0["losses"], r1["losses"], single["losses"]):
        assert abs(0.5 * (la + lb) - ls) <= 2e-4 * abs(ls) + 1e-5


def test_bench_entry_point_runs_two_ranks():
    """``python bench.py --gpus 2`` as the driver calls it (no torchrun environment): bench.py starts both ranks, the
    line reports ``n_gpus == rccl_world_size == 2``, consistent replicas, a measured all-reduce and twice the batch."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "2",
           "--size", "64", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_world_size"] == 2 and out["config"]["global_batch"] == 4
    assert out["config"]["dp_consistent"] is True and out["config"]["allreduce_ms_per_step"] > 0
    assert out["config"]["allreduce_overlap_frac"] is not None and out["value"] > 0
