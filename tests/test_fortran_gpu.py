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
    # numbers, not only "OK": the third sample (RHS + 50 PCG iterations + applyMonoDipolePrior) against the CPU oracle's
    # solution of the same LCG draws (tests/golden/mini_commander.json <- tests/golden/make_golden.py mini)
    import json
    import re
    import numpy as np
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "mini_commander.json")))
    num = r"[-+]?\d\.\d+E[-+]\d+"
    mu = [float(v) for v in re.findall(num, re.search(r"golden mu   =(.*)", out.stdout).group(1))]
    norm = float(re.search(r"golden norm =\s*(%s)" % num, out.stdout).group(1))
    amp = [float(re.search(r"golden amp%d =\s*(%s)" % (i, num), out.stdout).group(1)) for i in range(1, 9)]
    assert len(mu) == 4
    # fixed_iter, 50 iterations: rel-L2 <= 1e-8 between GPU and oracle; Box-Muller through two libm's
    # (measured: 1e-15 on a 10-iteration solve; the margin is for the two Box-Muller libm's and 50 iterations)
    assert abs(norm - gold["norm"]) < 1e-9 * gold["norm"]
    assert np.allclose(amp, gold["amp_first8"], rtol=1e-8, atol=1e-9)
    assert np.allclose(mu, gold["mu"], rtol=1e-7, atol=1e-10)


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
