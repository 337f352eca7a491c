"""GPU tier: the Fortran driver (ISO_C_BINDING -> libcmdr_hip.so) runs a few Gibbs amplitude samples."""
import os
import subprocess

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_mini_commander_runs():
    exe = os.path.join(ROOT, "fortran", "mini_commander")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/bin/amdflang"):
            pytest.skip("no Fortran compiler")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fortran")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mini_commander: OK" in out.stdout
    assert out.stdout.count("CG iters =  50") == 3


def test_api_tour_runs():
    """Every other part of the C ABI (mixing maps, compact components, error path, getSigmaL, chain order) called from
    Fortran through the ISO_C_BINDING module."""
    exe = os.path.join(ROOT, "fortran", "api_tour")
    if not os.path.exists(exe):
        if not os.path.exists("/opt/rocm/bin/amdflang"):
            pytest.skip("no Fortran compiler")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "fortran")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "api_tour: OK" in out.stdout
