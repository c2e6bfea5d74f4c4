import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure: build it on demand (gcc is on every box)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)


@pytest.fixture(scope="session")
def gpu():
    """Bind the engine to device 0; GPU tests fail loudly when the library or device is missing."""
    import dpx_gpu_genomics_project_amd as dpx

    dpx.init(0)
    return dpx
