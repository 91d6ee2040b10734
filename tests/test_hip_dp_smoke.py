"""-m gpu: the data-parallel machinery on a real GPU with RCCL -- torchrun, one rank, VUNET_DP_FORCE=1 makes the
averager register its hooks and issue the asynchronous bucket all-reduces exactly as with N ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_under_torchrun_with_forced_dp_path():
    env = dict(os.environ, VUNET_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "2", "--batch", "2", "--size", "64", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["config"]["final_loss"] == out["config"]["final_loss"]
    assert out["config"]["rccl_world_size"] == 1 and out["config"]["allreduce_ms_per_step"] > 0   # events on the comm stream
    assert out["config"]["dp_backend"] == "rccl-cabi"      # the step's all-reduces went through vunet_dp_allreduce_bucket


def test_bench_entry_point_starts_its_own_ranks():
    """``VUNET_DP_FORCE=1 python bench.py --gpus 1`` with NO torchrun environment: bench.py itself starts the rank
    process tree (a fresh child, before this parent touches the GPU), relays rank 0's ONE JSON line on stdout and its
    exit code -- the route `python bench.py --gpus N` takes for N > 1, exercised on the 1-GPU pool."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VUNET_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--batch", "2",
           "--size", "64", "--no-cpu-baseline", "--no-roofline", "--no-render", "--no-config1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                       # stdout carries the result line and nothing else
    out = json.loads(lines[0])
    assert "torch.distributed.run" in r.stderr                     # ... and it came from the rank tree bench.py started
    assert out["n_gpus"] == 1 and out["config"]["rccl_world_size"] == 1 and out["config"]["dp_consistent"] is True
    assert out["config"]["allreduce_ms_per_step"] > 0 and out["value"] > 0


def test_forced_one_rank_rccl_run_equals_the_plain_run(tmp_path):
    """tools/dp_check.py with one rank under torchrun (hooks, communication stream, RCCL all-reduce of every bucket, the
    averaged KL scalar) against the same steps without a process group: a one-rank SUM / 1 changes nothing, so the
    parameters must come out bit-identical -- any ordering bug between the weight-gradient companion streams, the
    communication stream and the optimiser's stream would show here."""
    import torch
    env = dict(os.environ, VUNET_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(ROOT, "tools", "dp_check.py")
    out = str(tmp_path)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", script, "--out", out, "--steps", "4"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    env.pop("VUNET_DP_FORCE")
    r = subprocess.run([sys.executable, script, "--single", "--world", "1", "--out", out, "--steps", "4"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = torch.load(os.path.join(out, "rank0.pt")), torch.load(os.path.join(out, "single.pt"))
    assert a["allreduce_ms"] is not None and b["allreduce_ms"] is None
    assert a["losses"] == b["losses"] and a["gamma"] == b["gamma"]
    for x, y in zip(a["flat"], b["flat"]):
        assert torch.equal(x, y)


def test_captured_step_with_rccl_allreduces_equals_the_eager_one(tmp_path):
    """The data-parallel step replayed from ONE hipGraph -- bucket all-reduces on the C-ABI RCCL communicator, communication
    stream forked and joined inside the capture -- against the same device-resident schedule launched eagerly: parameters,
    losses and gamma bit-identical after 9 steps (the capture happens at step 4, five replays follow)."""
    import torch
    env = dict(os.environ, VUNET_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(ROOT, "tools", "dp_check.py")
    res = {}
    for mode, port in (("capture", "29523"), ("eager", "29525")):
        out = str(tmp_path / mode)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", port, script, "--out", out, "--steps", "9",
                            "--graph", mode], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[mode] = torch.load(os.path.join(out, "rank0.pt"))
    a, b = res["capture"], res["eager"]
    assert a["backend"] == b["backend"] == "rccl-cabi" and a["graphs"] == 1 and b["graphs"] == 0
    assert a["losses"] == b["losses"] and a["gamma"] == b["gamma"]
    assert len(set(a["losses"])) == len(a["losses"])
    for x, y in zip(a["flat"], b["flat"]):
        assert torch.equal(x, y)


def test_failed_capture_with_an_active_averager_falls_back_to_the_eager_step(tmp_path):
    """a capture that fails AFTER the recording pass has run the averager's hooks and advanced the optimisers'
    host-side step counts.  The trainer must restart the averager's step state before it re-runs the step eagerly (or no
    all-reduce is issued and ``finish()`` refuses the step) and restore the counts: the run then equals the eager
    device-schedule run bit for bit, with no graph kept."""
    import torch
    env = dict(os.environ, VUNET_DP_FORCE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    script = os.path.join(ROOT, "tools", "dp_check.py")
    res = {}
    for mode, port in (("capture-fail", "29527"), ("eager", "29529")):
        out = str(tmp_path / mode)
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                            "--master-addr", "127.0.0.1", "--master-port", port, script, "--out", out, "--steps", "7",
                            "--graph", mode], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        if mode == "capture-fail":
            assert "injected capture failure" in r.stderr and "issued eagerly" in r.stderr.replace("\n", " ")
        res[mode] = torch.load(os.path.join(out, "rank0.pt"))
    a, b = res["capture-fail"], res["eager"]
    assert a["backend"] == b["backend"] == "rccl-cabi" and a["graphs"] == b["graphs"] == 0
    assert a["losses"] == b["losses"] and a["gamma"] == b["gamma"]
    assert a["adam_steps"] == b["adam_steps"] == [7, 7, 7, 7]
    for x, y in zip(a["flat"], b["flat"]):
        assert torch.equal(x, y)
