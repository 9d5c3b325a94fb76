import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _poison_uninitialised_memory():
    """SIMHAND_POISON=<GiB>: every torch.empty of the session returns NaN patterns (tests/_poison.py); the worker processes of the
    multi-rank tests do the same with SIMHAND_POISON_WORKER GiB each."""
    if os.environ.get("SIMHAND_POISON"):
        from tests._poison import poison

        poison()
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_hooks_back_to_default(request):
    """The library's test / tuning hooks (kernel-route switches) are process-global: whatever a test does with them, and
    however it ends, the next test starts from the defaults."""
    yield
    if "gpu" in request.keywords:
        import torch

        if torch.cuda.is_available():
            from simhand_amd import _lib, ops

            ops.hooks_reset()
            if _lib.half_format() != "bf16":  # a test that worked with the fp16 build: back to the default build (and its hooks)
                ops.hooks_reset()
                _lib.use_half("bf16")
