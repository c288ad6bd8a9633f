"""The C++ drop-in shim (include/edyn/edyn.hpp: edyn::attach / make_rigidbody / make_constraint / update over an
EnTT-compatible registry) builds with plain g++ against libedynhip.so, and - on a GPU - runs the reference's
README main loop."""
import os
import subprocess
import pytest
from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def test_shim_compiles_and_links():
    subprocess.check_call(["make", "-s", "-C", CPP, "hello_world"])
    assert os.path.exists(os.path.join(CPP, "hello_world"))


@pytest.mark.gpu
def test_shim_hello_world_runs():
    subprocess.check_call(["make", "-s", "-C", CPP, "hello_world"])
    out = subprocess.run([os.path.join(CPP, "hello_world")], capture_output=True, text=True, timeout=300)
    assert "HELLO_WORLD_OK" in out.stdout, out.stdout + out.stderr
