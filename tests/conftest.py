import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU-side libraries are cheap to build; the HIP library is built by __graft_entry__.build().
    host = os.path.join(ROOT, "raytracing-in-one-weekend_amd", "lib", "librtiow_host.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(host):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "raytracing-in-one-weekend_amd", "csrc"), "host"])
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def rt():
    import rtiow_loader
    return rtiow_loader.load()


@pytest.fixture(scope="session")
def gpu(rt):
    """One HipBackend for the whole session (tests run in one process)."""
    b = rt.backend.HipBackend(0)
    yield b
    b.close()
