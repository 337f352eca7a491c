"""Shared test helpers: build the numpy oracle system from a commander_amd.synth problem spec, load the host
emulation of libcmdr_hip (CPU-only tier)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def oracle_system(spec, only_pol=False):
    from oracle import cr_oracle as cro
    bands = [cro.Band(b["nside"], b["lmax"], b["siN"], b["b_l"], b.get("mb_eff", 1.0), b.get("sg_mask"), b.get("wring"))
             for b in spec["bands"]]
    comps = []
    for c in spec["comps"]:
        if c.get("sqrtS_mat") is None:
            cl = cro.Cl(c["lmax"], c["nmaps"], np.zeros((c["lmax"] + 1, c["nmaps"] * (c["nmaps"] + 1) // 2)), cltype="none")
        else:
            cl = cro.Cl(c["lmax"], c["nmaps"], c["Dl"])
            # the product receives the tables from commander_amd.cl.update_S; make sure both sides hold the same
            assert np.allclose(cl.sqrtS_mat, c["sqrtS_mat"], rtol=1e-13, atol=0)
        comps.append(cro.DiffuseComp(c["lmax"], c["nmaps"], cl, c["F_mean"], active=c.get("active", True),
                                     F_map=c.get("F_map")))
    return cro.CRSystem(bands, comps, only_pol=only_pol)


def emul_lib():
    """Host emulation of the library (tests/host_emul): same C ABI, kernels run as single-thread loops."""
    import importlib
    cl = importlib.import_module("commander_amd.lib")
    so = os.path.join(ROOT, "tests", "host_emul", "_build", "libcmdr_emul.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "commander_amd", "csrc"), "emul"],
                          stdout=subprocess.DEVNULL)
    return cl.load(so)
