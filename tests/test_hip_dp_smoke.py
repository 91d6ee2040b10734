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
