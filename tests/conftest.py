import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X; runs the HIP kernels through the C ABI")


@pytest.fixture(scope="session")
def default_scene():
    import raytracing_amd as rt

    return rt.Scene.default()


@pytest.fixture(scope="session")
def rtiow_scene():
    import raytracing_amd as rt

    return rt.Scene.generate(rt.KIND_RTIOW, 0, 7)
