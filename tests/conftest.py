import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "imm-tsf_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fresh_immtsf_state():
    """process-global switches of the library must not leak from one test into the next: the device-side dropout / step
    counters (enabled by FlatTrainer(device_step=True), bumped by every GraphedStep replay) and the precision mode"""
    yield
    try:
        from immtsf import config
    except Exception:       # noqa: BLE001  (CPU-only collection without the package importable)
        return
    config.disable_device_counters()
    config.precision = "fp32"
    config.nan_check = os.environ.get("IMMTSF_NAN_CHECK", "deferred")
