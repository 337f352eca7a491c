"""Shared test helpers: build the numpy oracle system from a commander_amd.synth problem spec, load the host
emulation of libcmdr_hip (CPU-only tier)."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / np.linalg.norm(b))


def oracle_system(spec, only_pol=False, literal_quirks=False):
    from oracle import cr_oracle as cro
    bands = [cro.Band(b["nside"], b["lmax"], b["siN"], b["b_l"], b.get("mb_eff", 1.0), b.get("sg_mask"), b.get("wring"))
             for b in spec["bands"]]
    for B, b in zip(bands, spec["bands"]):
        if b.get("qucov_iN") is not None:
            B.set_qucov(b["qucov_iN"], b["qucov_siN"])
    comps = []
    for c in spec["comps"]:
        if c.get("kind") == "compact":
            comps.append(cro.CompactBlock(c["nparam"], c["sigma"], c["mean"], c["P"], active=c.get("active", True)))
            continue
        if c.get("sqrtS_mat") is None:
            cl = cro.Cl(c["lmax"], c["nmaps"], np.zeros((c["lmax"] + 1, c["nmaps"] * (c["nmaps"] + 1) // 2)), cltype="none")
        else:
            cl = cro.Cl(c["lmax"], c["nmaps"], c["Dl"], l_apod=c.get("l_apod", 0), lmax_prior=c.get("lmax_prior", -1))
            # the product receives the tables from commander_amd.cl.update_S; make sure both sides hold the same
            assert np.allclose(cl.sqrtS_mat, c["sqrtS_mat"], rtol=1e-13, atol=0)
        comps.append(cro.DiffuseComp(c["lmax"], c["nmaps"], cl, c["F_mean"], active=c.get("active", True),
                                     F_map=c.get("F_map")))
    return cro.CRSystem(bands, comps, only_pol=only_pol, literal_quirks=literal_quirks)


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
    # ---- COMP_PRIOR_AMP_LMAX: the cosine roll-off of get_Cl_apod below lmax_prior on one component (T and T,Q,U)
    for pol in (False, True):
        spec = synth.make_problem("cfg2", nside=8, lmax=16, pol=pol)
        spec["comps"][1]["lmax_prior"] = 10
        S = oracle_system(spec)
        assert S.comps[1].Cl.f_apod[3] < 0.5 and S.comps[1].Cl.f_apod[10] == 1.0
        ctx = build_context(spec, _lib=_lib)
        x = rng.standard_normal(ctx.ncr)
        assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < tol
        resid, xi, eta = synth.draw_inputs(spec)
        b = S.computeRHS(resid, "sample", xi, eta)
        assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), b) < tol
        ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
        assert rel(ctx.cr_invM(x), S.invM(x)) < tol
        xg, ng, sg, _ = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=8)
        xo, no, so = S.solve(b, "fixed_iter", maxiter=8)
        assert rel(xg, xo) < 1e-9


def golden_checks(_lib=None, tol=1e-12):
    """The product against the committed golden vectors directly (tests/golden, independent of oracle/): brute-force
    spherical-harmonic sums for all four SHT jobs (spin 0, Nside 4 and 8) and the spin-2 pair (Nside 4), and the
    literal Wigner-3j evaluation of compute_invN_lm through the CR-level preconditioner setup."""
    import commander_amd.sht as shtmod
    from commander_amd.sht import ShtPlan
    from commander_amd.cr import CRContext
    G = os.path.join(ROOT, "tests", "golden")
    old = shtmod.lib
    if _lib is not None:
        shtmod.lib = lambda: _lib
    try:
        for nside in (4, 8):
            g = np.load(os.path.join(G, "sht_bruteforce_nside%d.npz" % nside))
            ns, lmax, w = int(g["nside"]), int(g["lmax"]), g["wring"]
            plan = ShtPlan(ns, lmax, wring=w, max_maps=1)
            assert rel(plan.Y(g["alm"][:, None])[:, 0], g["Y"]) < tol
            assert rel(plan.Yt(g["map"][:, None])[:, 0], g["Yt"]) < tol
            assert rel(plan.YtW(g["map"][:, None])[:, 0], g["YtW"]) < tol
            assert rel(plan.WY(g["alm"][:, None])[:, 0], g["WY"]) < tol
        g = np.load(os.path.join(G, "sht_spin2_bruteforce_nside4.npz"))
        ns, lmax = int(g["nside"]), int(g["lmax"])
        plan = ShtPlan(ns, lmax, max_maps=2, pol=True)
        q, u = plan.execute_spin2(1, almE=g["almE"], almB=g["almB"])
        assert rel(np.concatenate([q, u]), np.concatenate([g["Y_Q"], g["Y_U"]])) < tol
        e, b = plan.execute_spin2(2, mapQ=g["mapQ"], mapU=g["mapU"])
        assert rel(np.concatenate([e, b]), np.concatenate([g["Yt_E"], g["Yt_B"]])) < tol
    finally:
        shtmod.lib = old
    # compute_invN_lm: noise map in, literal 3j evaluation of the diagonal out (initDiffPrecond_diagonal's first half)
    g = np.load(os.path.join(G, "invn_diag_3j_map.npz"))
    ns, lmax = int(g["nside"]), int(g["lmax"])
    ctx = CRContext(0, _lib=_lib)
    ctx.add_band(ns, lmax, np.sqrt(g["siN2"]), np.ones(lmax + 1))
    ctx.add_comp(lmax, 1, np.ones((1, 1)), None, None, None, True)
    ctx.finalize()
    ctx.initPrecond()
    assert rel(ctx.invN_diag(0)[:, 0], g["diag"]) < 1e-11


def golden_kat(_lib=None):
    """The reference's only known-answer test (todscripts/wmap/cg_solver.py:54-61: A = [[3,2],[2,6]], b = [2,-8],
    M = I  =>  x = [2,-2]) has no 2x2 realisation in the CR operator; its recurrence is pinned on the oracle
    (tests/test_oracle.py) and the device PCG is pinned against the oracle's recurrence iteration by iteration."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg1", nside=4, lmax=6)
    S = oracle_system(spec)
    ctx = build_context(spec, _lib=_lib)
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    resid, xi, eta = synth.draw_inputs(spec)
    b = S.computeRHS(resid, "sample", xi, eta)
    for nit in (1, 2, 3, 7):
        xg, ng, sg, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=nit)
        hist = []
        xo, no, so = S.solve(b, "fixed_iter", maxiter=nit, history=hist)
        assert ng == no == nit and rel(xg, xo) < 1e-10
        assert abs(res[0] - hist[-1]) <= 1e-9 * abs(hist[-1])


def gibbs_loop_checks(_lib=None, nside=16, lmax=32, tol=1e-7):
    """Two Gibbs iterations of amplitudes | C_l and C_l | amplitudes through the product's update path
    (getSigmaL -> sampleCls_binned -> updateS -> set_comp_cl -> update_precond -> next solve), against the oracle
    rebuilt from scratch with the oracle's own sampled spectrum; then the sampling-group / mixing updates
    (set_active, set_comp_f_mean) against freshly built oracle systems.  Shared by the emulation and the GPU tier."""
    import copy
    from commander_amd import synth
    from commander_amd.cr import build_context, getSigmaL, sampleCls_binned, updateS
    from oracle import cl_oracle, cr_oracle
    spec = synth.make_problem("cfg2", nside=nside, lmax=lmax)
    ctx = build_context(spec, _lib=_lib)
    ctx.initPrecond()
    rng = np.random.default_rng(2024)
    RJ = np.ones(1)
    edges = [(2, 3), (4, 7), (8, 15), (16, lmax)]
    ospec = copy.deepcopy(spec)
    na0 = (lmax + 1) ** 2
    for it in range(2):
        S = oracle_system(ospec)
        S.init_precond_diag(); S.update_precond_diag()
        ctx.update_precond()
        resid, xi, eta = synth.draw_inputs(spec)
        resid = [r * (1.0 + 0.1 * it) for r in resid]
        cols = lambda lst: [np.asarray(v)[:, None] for v in lst]  # noqa: E731
        rhs = ctx.cr_computeRHS("sample", resid, xi, eta)
        rhso = S.computeRHS(cols(resid), "sample", cols(xi), eta)
        # second pass: the two sides carry their own sampled spectrum (equal to ~tol), so the prior term differs at that level
        assert rel(rhs, rhso) < (1e-11 if it == 0 else 10 * tol), (it, rel(rhs, rhso))
        x, n, stat, res = ctx.solve_cr_eqn_by_CG(rhs, "residual", 1e-12, 5, 600, 1)
        xo, no, so = S.solve(rhso, "residual", 1e-12, 5, 600, 1)
        assert rel(x, xo) < (tol if it == 0 else 10 * tol), (it, rel(x, xo))
        # C_l | a_lm for the CMB component (first block of the stacked vector)
        Dl = np.asarray(spec["comps"][0]["Dl"], dtype=np.float64).reshape(lmax + 1, 1)
        Dlo = np.asarray(ospec["comps"][0]["Dl"], dtype=np.float64).reshape(lmax + 1, 1)
        bins = [dict(lmin=a, lmax=b, spec=1, sample=True, sigma=0.1 * Dl[a, 0]) for a, b in edges]
        u = rng.uniform(size=len(bins))
        sig = getSigmaL(x[:na0], lmax, _lib=_lib)
        sigo = cr_oracle.getSigmaL(xo[:na0], lmax)
        assert rel(sig, sigo) < tol
        newDl, ok, used = sampleCls_binned(Dl, sig, spec["comps"][0]["S_mat"], RJ, bins, u, _lib=_lib)
        newDlo = Dlo.copy()
        oko, usedo = cl_oracle.sample_cls_binned(newDlo, sigo, ospec["comps"][0]["S_mat"], RJ, bins, u)
        assert ok and oko and used == usedo == len(bins)
        assert rel(newDl, newDlo) < 10 * tol, (it, rel(newDl, newDlo))
        assert rel(newDl, Dl) > 1e-3                      # it did move
        sq, isq, Sm, nbad = updateS(newDl, 0, RJ, _lib=_lib)
        assert nbad == 0
        ctx.set_comp_cl(0, sq, isq, Sm)
        spec["comps"][0].update(Dl=newDl[:, 0], sqrtS_mat=sq, sqrtInvS_mat=isq, S_mat=Sm)
        o = cl_oracle.update_S(newDlo, 0, RJ)
        ospec["comps"][0].update(Dl=newDlo[:, 0], sqrtS_mat=o[0], sqrtInvS_mat=o[1], S_mat=o[2])
    # the matvec sees the new prior at once
    xt = rng.standard_normal(ctx.ncr)
    S = oracle_system(ospec)
    assert rel(ctx.cr_matmulA(xt), S.matmulA(xt)) < 10 * tol
    # ---- next sampling group: second component switched off, then on again with new mixing
    ospec["comps"][1]["active"] = False
    ctx.set_active(1, False)
    S = oracle_system(ospec)
    y = ctx.cr_matmulA(xt)
    assert rel(y, S.matmulA(xt)) < 10 * tol and np.all(y[na0:] == 0.0)
    ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(xt), S.invM(xt)) < 10 * tol
    ospec["comps"][1]["active"] = True
    ospec["comps"][1]["F_mean"] = ospec["comps"][1]["F_mean"] * np.array([1.3, 1.0, 0.6])[:, None]
    ctx.set_active(1, True)
    ctx.set_comp_f_mean(1, ospec["comps"][1]["F_mean"])
    S = oracle_system(ospec)
    assert rel(ctx.cr_matmulA(xt), S.matmulA(xt)) < 10 * tol
    # new mixing: the reference sets recompute_diffuse_precond and runs initPrecond again before the next solve
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(xt), S.invM(xt)) < 10 * tol


def residual_checks(_lib=None, nside=8, lmax=16, tol=1e-11):
    """compute_residual(cg_samp_group) through the product against the oracle: components outside the sampling group
    (constant mixing, spatially varying mixing, a template + point-source block) are subtracted from the data; the
    group's own components are not; afterwards the solver context is back in its state (same matvec)."""
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    rng = np.random.default_rng(31)
    for pol in (False, True):
        spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, pol=pol, comp_lmax=[lmax, lmax - 4])
        nm = 3 if pol else 1
        z = healpix.pix_z(nside)
        spec["comps"][1]["F_map"] = {ib: np.repeat(((b["nu"] / 30.0) ** (-3.1 + 0.1 * z))[:, None], nm, axis=1)
                                     for ib, b in enumerate(spec["bands"])}
        synth.add_compact_blocks(spec, nsrc=3)
        # sampling group = CMB only: synchrotron (varying mixing) and the compact blocks are "the rest of the sky model"
        for c in spec["comps"][1:]:
            c["active"] = False
        S = oracle_system(spec)
        ctx = build_context(spec, _lib=_lib)
        amp = rng.standard_normal(ctx.ncr)
        data = [rng.standard_normal(b["siN"].shape if pol else (len(b["siN"]), 1)) for b in spec["bands"]]
        x = rng.standard_normal(ctx.ncr)
        y0 = ctx.cr_matmulA(x)
        got = ctx.compute_residual(amp, data)
        want = S.compute_residual(data, amp)
        for g, w, d in zip(got, want, data):
            assert rel(g, w) < tol, (pol, rel(g, w))
            assert rel(g, d.reshape(g.shape)) > 1e-3           # something was subtracted
        assert np.array_equal(ctx.cr_matmulA(x), y0)            # flags, weights and mixing batches restored
        # nothing outside the group -> the data come back unchanged
        for c in spec["comps"]:
            c["active"] = True
        ctx2 = build_context(spec, _lib=_lib)
        for g, d in zip(ctx2.compute_residual(amp, data), data):
            assert np.array_equal(g, d.reshape(g.shape))


def chisq_criterion_checks(_lib=None, nside=8, lmax=16):
    """The 'chisq' convergence criterion (cr_compute_chisq, comm_cr_mod.f90:223-242, 408-465): same stopping iteration
    and solution as the oracle, T and T,Q,U, with a samp-group mask (chisq is evaluated without it), a component outside
    the sampling group folded into the residual, and a compact block inside it."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    rng = np.random.default_rng(41)
    for pol in (False, True):
        spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, pol=pol)
        synth.add_compact_blocks(spec, nsrc=2)
        for b in spec["bands"]:
            b["sg_mask"] = (rng.random(b["siN"].shape) > 0.2).astype(np.float64)
        S = oracle_system(spec)
        ctx = build_context(spec, _lib=_lib)
        ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
        resid, xi, eta = synth.draw_inputs(spec)
        shp = [np.asarray(b["siN"]).reshape(len(b["siN"]), -1).shape for b in spec["bands"]]
        rcols = [np.asarray(r, dtype=np.float64).reshape(s) for r, s in zip(resid, shp)]
        xcols = [np.asarray(r, dtype=np.float64).reshape(s) for r, s in zip(xi, shp)]
        rhs = ctx.cr_computeRHS("sample", resid, xi, eta)
        rhso = S.computeRHS(rcols, "sample", xcols, eta)
        assert rel(rhs, rhso) < 1e-11
        # limits well above the level where rounding differences between two fp64 CG trajectories decide the test
        for tol_c, freq in ((1e-2, 1), (1e-4, 2)):
            xg, ng, sg, _ = ctx.solve_cr_eqn_by_CG(rhs, "chisq", tol_c, 2, 200, freq)
            xo, no, so = S.solve(rhso, "chisq", tol_c, 2, 200, freq, resid=rcols)
            assert ng == no and sg == so and 2 <= ng < 200, (pol, tol_c, ng, no)
            # CG trajectories of two fp64 implementations drift apart at rounding level; the stopping iteration is the check
            assert rel(xg, xo) < 1e-5, (pol, tol_c, rel(xg, xo))


def pol_pruned_checks(_lib=None, nside=256, lmax=512, tol=1e-11):
    """Polarised CR operator at a size where the (m, ring) pruning is active and differs between spin 0 and spin 2
    (Nside >= 256 / lmax 512), with azimuth-dependent noise and mask so that every m couples to every other.
    Regression for the stale-phase bug of round 1 (T slots of a polarised plan kept analysis output in the entries
    mlim_spin0 < m <= mlim_spin2): A x must repeat bit-identically whatever ran in between, and the later calls must
    match the oracle like the first."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg4", nside=nside, lmax=lmax, aniso=0.3)
    ctx = build_context(spec, _lib=_lib)
    S = oracle_system(spec)
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Ax1 = ctx.cr_matmulA(x)
    Ay = ctx.cr_matmulA(y)
    Ax3 = ctx.cr_matmulA(x)
    assert np.array_equal(Ax1, Ax3), rel(Ax1, Ax3)
    assert rel(Ax1, S.matmulA(x)) < tol
    assert rel(Ay, S.matmulA(y)) < tol                     # the second call, after a first one left its phases behind
    assert abs(y @ Ax1 - x @ Ay) < tol * abs(y @ Ax1)      # symmetry
    resid, xi, eta = synth.draw_inputs(spec)
    rhs = ctx.cr_computeRHS("sample", resid, xi, eta)
    assert rel(rhs, S.computeRHS(resid, "sample", xi, eta)) < tol
    assert np.array_equal(ctx.cr_matmulA(x), Ax1)          # ... and after an RHS (analysis-only calls)
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(x), S.invM(x)) < tol
    assert np.array_equal(ctx.cr_matmulA(x), Ax1)


def lowl_precond_checks(_lib=None, nside=16, lmax=32, L=6, nside_low=4, tol=1e-10):
    """CG_LMAX_PRECOND: the low-l dense preconditioner block (updateLowlPrecond / applyLowlPrecond,
    comm_diffuse_comp_mod.f90:5098-5310; cr_invM, comm_cr_mod.f90:1058-1073) against the oracle: cr_invM itself, a
    fixed_iter solve through it, the rebuild after new C_l, and switching it off again."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    for pol in (False, True):
        spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, pol=pol, aniso=0.3)
        low = synth.lowres_noise(spec, nside_low)
        S = oracle_system(spec)
        for B, (ns, m) in zip(S.bands, low):
            B.set_lowres(ns, m)
        ctx = build_context(spec, _lib=_lib)
        ctx.initPrecond(); ctx.update_precond()
        S.init_precond_diag(); S.update_precond_diag()
        x = np.random.default_rng(8).standard_normal(ctx.ncr)
        plain = ctx.cr_invM(x)
        ctx.set_lowl_precond(0, L, [ns for ns, _ in low], [m for _, m in low])
        ctx.update_precond()
        S.set_lowl(0, L); S.update_lowl()
        got, want = ctx.cr_invM(x), S.invM(x)
        assert rel(got, want) < tol, rel(got, want)
        n_low = (L + 1) ** 2
        assert np.count_nonzero(got != plain) == n_low            # exactly the T entries with l <= L changed
        resid, xi, eta = synth.draw_inputs(spec)
        b = S.computeRHS(resid, "sample", xi, eta)
        xg, ng, _, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=12)
        xo, no, _ = S.solve(b, "fixed_iter", maxiter=12)
        assert ng == no == 12 and rel(xg, xo) < 1e-8
        ctx.set_lowl_precond(0, -1)
        assert np.array_equal(ctx.cr_invM(x), plain)


def mono_dipole_prior_checks(_lib=None, tol=1e-11):
    """``applyMonoDipolePrior`` (comm_diffuse_comp_mod.f90:5738-5827), the tail of ``sample_amps_by_CG``
    (comm_signal_mod.f90:186-194): product vs oracle for 'monopole' and 'monopole+dipole', with and without an output
    beam, T and T,Q,U components, the component on a band plan's geometry and on one of its own (other lmax, other
    nside) -- and, independent of the reference's text, the property that pins the four sign / normalisation factors of
    :5811-5824: without an output beam the masked fit of the corrected map is zero."""
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    from oracle import sht as osht, healpix as ohp
    rng = np.random.default_rng(31)
    for pol, comp_lmax, ns_c in ((False, [32, 24], 16), (True, [32, 32], 16), (False, [20, 32], 8)):
        spec = synth.make_problem("cfg2", nside=16, lmax=32, comp_lmax=comp_lmax, pol=pol)
        S = oracle_system(spec)
        ctx = build_context(spec, _lib=_lib)
        amp = rng.standard_normal(ctx.ncr) * 10.0
        z = healpix.pix_z(ns_c)
        mask = ((np.abs(z) > 0.3) & (rng.random(z.size) > 0.1)).astype(np.float64)
        wmask = mask * (0.6 + 0.4 * rng.random(z.size))             # 'monopole' uses the mask as a weight (:5764-5765)
        for k in range(2):
            lm = spec["comps"][k]["lmax"]
            bl = np.exp(-0.5 * np.arange(lm + 1) * (np.arange(lm + 1) + 1.0) * (np.radians(1.5) / 2.355) ** 2)
            for ptype, m in (("monopole", wmask), ("monopole+dipole", wmask)):
                for b_l_out in (None, bl):
                    got, mu = ctx.applyMonoDipolePrior(k, amp, ns_c, m, ptype, b_l_out)
                    want, muo = S.apply_mono_dipole_prior(k, amp, ns_c, m, ptype, b_l_out)
                    assert rel(got, want) < tol, (pol, k, ptype, rel(got, want))
                    assert np.allclose(mu, muo, rtol=1e-9, atol=1e-11 * np.abs(muo).max()), (mu, muo)
                    changed = np.flatnonzero(got != amp)
                    assert 1 <= changed.size <= (1 if ptype == "monopole" else 4)      # only (0,0), (1,-1), (1,0), (1,1)
                    if b_l_out is None:      # the corrected map has no masked monopole (/ dipole) left
                        _, mu2 = S.apply_mono_dipole_prior(k, got, ns_c, m, ptype, None)
                        assert np.abs(mu2).max() < 1e-10 * max(np.abs(muo).max(), 1.0), mu2
        # device-resident form == host form
        if hasattr(ctx, "dev"):
            d_amp, d_mask = ctx.dev(ctx.ncr, amp), ctx.dev(wmask.size, wmask)
            mu_d = ctx.applyMonoDipolePrior_dev(0, d_amp, ns_c, d_mask, "monopole+dipole")
            ref, mu_h = ctx.applyMonoDipolePrior(0, amp, ns_c, wmask, "monopole+dipole")
            assert np.array_equal(d_amp.download(), ref) and np.array_equal(mu_d, mu_h)
    # the dipole of a pure (1, m) map in HEALPix vector components: unit x, y, z dipoles come back as mu = e_x, e_y, e_z
    nside, lmax = 8, 4
    th, ph = ohp.pix_angles(nside)
    spec = synth.make_problem("cfg1", nside=nside, lmax=lmax)
    ctx = build_context(spec, _lib=_lib)
    info = ohp.AlmInfo(lmax)
    s1 = np.sqrt(4.0 * np.pi / 3.0)
    for j, vec in enumerate((np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th))):
        a = np.zeros(info.nalm)
        a[info.lm2i(0, 0)] = 3.0 * np.sqrt(4.0 * np.pi)
        a[info.lm2i(1, (1, -1, 0)[j])] = 2.0 * s1 * (-1.0, 1.0, 1.0)[j]
        assert np.allclose(osht.Y(nside, lmax, a), 3.0 + 2.0 * vec, atol=1e-12)      # the SHT is pinned by brute force
        got, mu = ctx.applyMonoDipolePrior(0, a, nside, np.ones(th.size), "monopole+dipole")
        e = np.zeros(4); e[0] = 3.0; e[1 + j] = 2.0
        assert np.allclose(mu, e, atol=1e-10), (j, mu)
        assert np.abs(got).max() < 1e-10                                             # nothing is left of the map


def literal_quirks_checks(_lib=None, nside=16, lmax=32, tol=1e-11):
    """cr_matmulA's literal buffer re-use (comm_cr_mod.f90:846-861): with components of different lmax_amp the later,
    smaller one reads the earlier one's coefficients above its own lmax.  Product and oracle agree in both modes, the
    modes differ, and only the literal one is non-symmetric (T and T,Q,U; three components with mixed lmax)."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    rng = np.random.default_rng(21)
    for pol in (False, True):
        cfg = dict(synth.CONFIGS["cfg2"], comps=["cmb", "synch", "dust"])
        spec = synth.make_problem(cfg, nside=nside, lmax=lmax, comp_lmax=[lmax, lmax - 8, lmax - 3], pol=pol)
        ctx = build_context(spec, _lib=_lib)
        x, y = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
        S0, S1 = oracle_system(spec), oracle_system(spec, literal_quirks=True)
        A0 = ctx.cr_matmulA(x)
        assert rel(A0, S0.matmulA(x)) < tol
        ctx.set_literal_quirks(True)
        A1, A1y = ctx.cr_matmulA(x), ctx.cr_matmulA(y)
        assert rel(A1, S1.matmulA(x)) < tol
        assert rel(A1, A0) > 1e-6                                        # the quirk is not a rounding effect
        assert abs(y @ A1 - x @ A1y) > 1e-8 * abs(y @ A1)                # ... and it breaks the symmetry of A
        assert abs(y @ A0 - x @ ctx_sym(ctx, y)) <= 1e-11 * abs(y @ A0)
        # the RHS and the preconditioner do not go through the re-used buffer
        resid, xi, eta = synth.draw_inputs(spec)
        assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), S1.computeRHS(resid, "sample", xi, eta)) < tol


def ctx_sym(ctx, v):
    ctx.set_literal_quirks(False)
    out = ctx.cr_matmulA(v)
    ctx.set_literal_quirks(True)
    return out


def pinv_toeplitz_checks(_lib=None, nside=256, lmax=512, tol=1e-10):
    """Pseudo-inverse preconditioner at a size where the polar-cap rings take the Toeplitz (circulant) ring form
    (Nside >= 256): applyDiffPrecond_pseudoinv's N operator (WY . N . YtW, comm_diffuse_comp_mod.f90:2293-2299) against
    the oracle, with azimuth-dependent noise."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, aniso=0.3)
    ctx = build_context(spec, _lib=_lib)
    S = oracle_system(spec)
    ctx.initPrecond("pseudoinv"); ctx.update_precond()
    S.init_precond_pseudoinv(); S.update_precond_pseudoinv()
    for b in range(len(spec["bands"])):
        assert abs(ctx.alpha_nu(b)[0] / S.bands[b].alpha_nu[0] - 1.0) < 1e-11
    x = np.random.default_rng(17).standard_normal(ctx.ncr)
    got = ctx.cr_invM(x)
    assert rel(got, S.invM(x)) < tol
    assert np.array_equal(ctx.cr_invM(x), got)


def fused_staging_checks(_lib=None, nside=128, lmax=24, tol=1e-12):
    """256 ring pairs: the synthesis runs in its workgroup form, where the coefficient stream is never written -- the
    kernel's tile staging forms beam x mixing x component sums itself (PrepDev), incl. the varying-mixing 'extra' term
    and components with different lmax; matvec, RHS and the pseudo-inverse preconditioner against the oracle."""
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, comp_lmax=[lmax, lmax - 7])
    z = healpix.pix_z(nside)
    spec["comps"][1]["F_map"] = {ib: ((b["nu"] / 30.0) ** (-3.1 + 0.1 * z))[:, None]
                                 for ib, b in enumerate(spec["bands"])}
    S = oracle_system(spec)
    ctx = build_context(spec, _lib=_lib)
    x = np.random.default_rng(33).standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < tol
    resid, xi, eta = synth.draw_inputs(spec)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), S.computeRHS(resid, "sample", xi, eta)) < tol
    ctx.initPrecond("pseudoinv")
    ctx.update_precond()
    S.init_precond_pseudoinv()
    S.update_precond_pseudoinv()
    assert rel(ctx.cr_invM(x), S.invM(x)) < 10 * tol


def fused_pcg_checks(_lib, pol, monkeypatch, nside=16, lmax=32, cfg="cfg2"):
    """solve_cr_eqn_by_CG with the three fused vector kernels per iteration (S^1/2 yc + d with d.q; x, r, M^-1 r with
    r.s; d with the next S^1/2 d) against the one-kernel-per-line sequence (CMDR_CG_FUSED=0) and the oracle, with
    different lmax per component and the chisq criterion (whose evaluation overwrites the S^1/2 buffer)."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    # cfg3 = nine bands, one T-only component: the kernels' single-component form; cfg2: the general one
    spec = synth.make_problem(cfg, nside=nside, lmax=lmax, pol=pol,
                              comp_lmax=[lmax, lmax - 8] if cfg == "cfg2" else None)
    S = oracle_system(spec)
    ctx = build_context(spec, _lib=_lib)
    ctx.initPrecond()
    ctx.update_precond()
    S.init_precond_diag()
    S.update_precond_diag()
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("CMDR_CG_FUSED", mode)
        out[mode] = [ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1),
                     ctx.solve_cr_eqn_by_CG(b, "residual", 1e-10, 5, 600, 3),
                     ctx.solve_cr_eqn_by_CG(b, "chisq", 1e-3, 2, 60, 2)]
    f, g = out["1"][0], out["0"][0]                                # fixed_iter: same arithmetic up to the dot-product order
    assert f[1] == g[1] == 12 and f[2] == g[2] and rel(f[0], g[0]) < 1e-9, (f[1:], g[1:], rel(f[0], g[0]))
    assert abs(f[3][0] - g[3][0]) <= 1e-7 * abs(g[3][0])           # delta_new
    # stopping rules: the two sequences sum their dot products in different orders, so after many iterations the
    # iterates differ at the level the stopping rule leaves (delta_new / delta_0 < 1e-10 -> ~1e-5 in the solution)
    for k, freq, lim in ((1, 3, 1e-4), (2, 2, 1e-6)):
        f, g = out["1"][k], out["0"][k]
        assert abs(f[1] - g[1]) <= freq and f[2] == g[2] == 0 and rel(f[0], g[0]) < lim, (k, f[1:], g[1:], rel(f[0], g[0]))
    xo, no, so = S.solve(b, "fixed_iter", 1e-8, 5, 12, 1)
    assert out["1"][0][1] == no and rel(out["1"][0][0], xo) < 1e-10
