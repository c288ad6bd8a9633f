import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = _have_gpu()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must not silently pass without a device: they fail loudly inside edynhip_create.
    pass


@pytest.fixture(scope="session", autouse=True)
def build_native():
    """Build the oracle (and, where /root/reference exists, the reference cross-check lib) once per session.
    The product library is built by __graft_entry__.build(); tests never rebuild it on the GPU box."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    if os.path.isdir("/root/reference/src/edyn") :
        subprocess.call(["make", "-s", "-j%d" % max(2, os.cpu_count() or 2), "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(os.path.join(ROOT, "edyn_amd", "libedynhip.so")):
        subprocess.check_call(["make", "-s", "-j4", "-C", os.path.join(ROOT, "edyn_amd", "csrc")])
    # the test meshes take the first ids of the process-wide mesh registries (tests/meshes.py), whatever test runs first
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import meshes
    meshes.registered()
    yield
