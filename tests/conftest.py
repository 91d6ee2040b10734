import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (meta dict, arrays dict) of tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["__meta__"]).decode())
    return meta, {k: z[k] for k in z.files if k != "__meta__"}


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _reset_library_tuning():
    """Dispatcher overrides set by a test (ops.set_tuning) never leak into the next one."""
    yield
    from behavior_driven_video_synthesis_amd import _lib
    if _lib._LIB is not None:
        from behavior_driven_video_synthesis_amd import ops
        for key in ops._TUNING_KEYS:
            ops.set_tuning(key, 0)
