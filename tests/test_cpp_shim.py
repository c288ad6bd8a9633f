"""The C++ drop-in shim (include/edyn/edyn.hpp: edyn::attach / make_rigidbody / make_constraint / update over an
EnTT-compatible registry) builds with plain g++ against libedynhip.so, and - on a GPU - runs the reference's
README main loop."""
import os
import subprocess
import pytest
from conftest import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")


def test_shim_compiles_and_links():
    subprocess.check_call(["make", "-s", "-C", CPP, "all"])
    assert os.path.exists(os.path.join(CPP, "hello_world")) and os.path.exists(os.path.join(CPP, "lifecycle"))


@pytest.mark.gpu
def test_shim_hello_world_runs():
    subprocess.check_call(["make", "-s", "-C", CPP, "hello_world"])
    out = subprocess.run([os.path.join(CPP, "hello_world")], capture_output=True, text=True, timeout=300)
    assert "HELLO_WORLD_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_lifecycle_runs():
    """registry.destroy on bodies and constraints, clear_rigidbody, runtime settings, constraint overloads with optional rows,
    exclude_collision, capacity growth that carries the contacts, update(registry) - tests/cpp/lifecycle.cpp."""
    subprocess.check_call(["make", "-s", "-C", CPP, "lifecycle"])
    out = subprocess.run([os.path.join(CPP, "lifecycle")], capture_output=True, text=True, timeout=300)
    assert "LIFECYCLE_OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_contact_entities_and_asynchronous_mode():
    """contact_manifold / contact_point entities mirrored from the device's event list; execution_mode::asynchronous delivers
    the synchronous trajectory one update late - tests/cpp/contacts.cpp."""
    subprocess.check_call(["make", "-s", "-C", CPP, "contacts"])
    out = subprocess.run([os.path.join(CPP, "contacts")], capture_output=True, text=True, timeout=300)
    assert "contacts OK" in out.stdout, out.stdout + out.stderr
