import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "auto_path: let the library pick the kernel path by batch size")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _wave_path_for_small_test_batches(request):
    """The GPU tests run batches of 1..70 scenes.  By default the library hands such small batches to the
    workgroup-per-scene kernels; the tests are there to pin the kernels the bench-sized batches run, so they keep the
    wave-per-scene path on (STG_OPT_WAVE_PATH) unless a test chooses a path itself (wg_path / `auto_path` marker)."""
    if "gpu" not in request.keywords:
        yield
        return
    from social_stgcnn_amd import ops
    old = dict(ops.OPTIONS)
    ops.OPTIONS["wave_path"] = "auto_path" not in request.keywords
    yield
    ops.OPTIONS.clear()
    ops.OPTIONS.update(old)
