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


def edge_case_checks(_lib=None, tol=1e-11):
    """Less-travelled branches of the CR path against the oracle, shared by the emulation (CPU) and GPU tiers:
    samp-group mask, mb_eff != 1, prior mean mu in the RHS, operation /= 'sample', only_pol, a component without
    prior, tiny geometries (Nside 1-2, lmax 0-2), band lmax above and below the component's."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    rng = np.random.default_rng(77)
    # ---- masks, mb_eff, mu, operation
    spec = synth.make_problem("cfg2", nside=8, lmax=16, comp_lmax=[16, 12])
    for ib, b in enumerate(spec["bands"]):
        b["mb_eff"] = 1.0 + 0.1 * ib
        b["sg_mask"] = (rng.random(b["siN"].shape) > 0.3).astype(np.float64)
    S = oracle_system(spec)
    ctx = build_context(spec, _lib=_lib)
    x = rng.standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < tol
    resid, xi, eta = synth.draw_inputs(spec)
    mu = rng.standard_normal(ctx.ncr)
    for k, c in enumerate(S.comps):
        c.mu = S.extract(k, mu)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta, mu=mu), S.computeRHS(resid, "sample", xi, eta)) < tol
    assert rel(ctx.cr_computeRHS("mean", resid, mu=mu), S.computeRHS(resid, "mean")) < tol
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(x), S.invM(x)) < tol
    # ---- only_pol on a polarised problem; second component without a prior (cltype 'none')
    spec = synth.make_problem("cfg2", nside=4, lmax=8, pol=True)
    for k in ("sqrtS_mat", "sqrtInvS_mat", "S_mat"):
        spec["comps"][1][k] = None
    S = oracle_system(spec, only_pol=True)
    ctx = build_context(spec, _lib=_lib)
    ctx.set_only_pol(True)
    x = rng.standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < tol
    resid, xi, eta = synth.draw_inputs(spec)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), S.computeRHS(resid, "sample", xi, eta)) < tol
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(x), S.invM(x)) < tol
    # ---- tiny geometries and mismatched lmax
    for nside, lmax, clm in [(1, 0, [0]), (1, 2, [1]), (2, 1, [4]), (2, 5, [2]), (4, 3, [9])]:
        spec = synth.make_problem("cfg1", nside=nside, lmax=lmax, comp_lmax=clm)
        S = oracle_system(spec)
        ctx = build_context(spec, _lib=_lib)
        x = rng.standard_normal(ctx.ncr)
        assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < tol, (nside, lmax, clm)
        resid, xi, eta = synth.draw_inputs(spec)
        assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), S.computeRHS(resid, "sample", xi, eta)) < tol
        ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
        b = S.computeRHS(resid, "sample", xi, eta)
        nit = min(4, ctx.ncr)      # CG on an n-dimensional system is exact after n steps; beyond that it is 0/0
        xg, ng, sg, _ = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=nit)
        xo, no, so = S.solve(b, "fixed_iter", maxiter=nit)
        assert rel(xg, xo) < 1e-9, (nside, lmax, clm)
