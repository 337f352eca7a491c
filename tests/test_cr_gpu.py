"""GPU parity of the CR path (cr_matmulA / cr_invM / cr_computeRHS / solve_cr_eqn_by_CG through the C ABI) against
the numpy+C oracle on identical seeded inputs.  Tolerances from BASELINE.md §4 / SURVEY.md §8c:
matvec <= 1e-11, fixed_iter solve <= 1e-8, converged solve <= 1e-6 with iteration count +-1."""
import numpy as np
import pytest

from helpers import oracle_system, rel

pytestmark = pytest.mark.gpu


def _cols(lst):
    return [np.asarray(v)[:, None] for v in lst]


@pytest.fixture(scope="module")
def small():
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=32, lmax=64, comp_lmax=[64, 48])
    ctx = build_context(spec)
    ctx.initPrecond()
    ctx.update_precond()
    S = oracle_system(spec)
    S.init_precond_diag()
    S.update_precond_diag()
    return spec, ctx, S


def test_matmulA_invM_small(small):
    spec, ctx, S = small
    rng = np.random.default_rng(11)
    for _ in range(2):
        x = rng.standard_normal(ctx.ncr)
        assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
        assert rel(ctx.cr_invM(x), S.invM(x)) < 1e-11
    for b in range(len(spec["bands"])):
        assert rel(ctx.invN_diag(b)[:, 0], S.bands[b].invN_diag[:, 0]) < 1e-11


def test_matmulA_symmetric_positive(small):
    spec, ctx, S = small
    rng = np.random.default_rng(12)
    u, v = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Au, Av = ctx.cr_matmulA(u), ctx.cr_matmulA(v)
    assert abs(v @ Au - u @ Av) <= 1e-11 * abs(v @ Au)
    assert u @ Au > u @ u  # A = 1 + (positive semi-definite)


def test_rhs_small(small):
    from commander_amd import synth
    spec, ctx, S = small
    resid, xi, eta = synth.draw_inputs(spec)
    rhs = ctx.cr_computeRHS("sample", resid, xi, eta)
    assert rel(rhs, S.computeRHS(_cols(resid), "sample", _cols(xi), eta)) < 1e-11
    rhs2 = ctx.cr_computeRHS("optimize", resid)
    assert rel(rhs2, S.computeRHS(_cols(resid), "optimize")) < 1e-11


def test_solve_fixed_iter_and_residual_small(small):
    from commander_amd import synth
    spec, ctx, S = small
    resid, xi, eta = synth.draw_inputs(spec)
    b = S.computeRHS(_cols(resid), "sample", _cols(xi), eta)
    x, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 40, 1)
    xo, no, so = S.solve(b, "fixed_iter", 1e-8, 5, 40, 1)
    assert n == no == 40 and stat == so == 0
    assert rel(x, xo) < 1e-8
    x, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "residual", 1e-4, 5, 200, 1)
    xo, no, so = S.solve(b, "residual", 1e-4, 5, 200, 1)
    assert abs(n - no) <= 1 and stat == so == 0
    assert rel(x, xo) < 1e-6
    # warm start from the previous solution (cg_init_zero = .false., comm_cr_mod.f90:136-173)
    x2, n2, stat2, _ = ctx.solve_cr_eqn_by_CG(b, "residual", 1e-4, 1, 200, 1, x0=x)
    xo2, no2, so2 = S.solve(b, "residual", 1e-4, 1, 200, 1, x0=xo)
    assert abs(n2 - no2) <= 1 and n2 < n
    assert rel(x2, xo2) < 1e-6


def test_cfg2_matvec_and_rhs_full_size():
    """BASELINE.json configs[1]: 3 bands, CMB+synch, Nside=256, lmax=512."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2")
    ctx = build_context(spec)
    S = oracle_system(spec)
    rng = np.random.default_rng(13)
    x = rng.standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
    resid, xi, eta = synth.draw_inputs(spec)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), S.computeRHS(_cols(resid), "sample", _cols(xi), eta)) < 1e-11
    ctx.initPrecond()
    ctx.update_precond()
    S.init_precond_diag()
    S.update_precond_diag()
    assert rel(ctx.cr_invM(x), S.invM(x)) < 1e-11
    b = S.computeRHS(_cols(resid), "sample", _cols(xi), eta)
    xg, n, stat, _ = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 10, 1)
    xo, no, so = S.solve(b, "fixed_iter", 1e-8, 5, 10, 1)
    assert n == no == 10
    assert rel(xg, xo) < 1e-8


def test_cfg3_full_size_properties():
    """BASELINE.json configs[2] geometry (9 bands, Nside=1024, lmax=2000): size-independent properties only --
    symmetry of A, A u > u, M^-1 positive, and PCG actually reducing r^T M^-1 r by the expected orders."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg3")
    ctx = build_context(spec)
    rng = np.random.default_rng(14)
    u, v = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Au, Av = ctx.cr_matmulA(u), ctx.cr_matmulA(v)
    assert abs(v @ Au - u @ Av) <= 1e-10 * abs(v @ Au)
    assert u @ Au > u @ u
    ctx.initPrecond()
    ctx.update_precond()
    Mu = ctx.cr_invM(u)
    assert u @ Mu > 0
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    x, n, stat, (dn, d0) = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 40, 1)
    assert n == 40 and np.isfinite(x).all()
    assert dn < 1e-3 * d0  # masked sky + diagonal preconditioner: ~4 orders in 40 iterations


def test_sigma_l_gpu():
    from commander_amd.cr import getSigmaL
    from oracle import cr_oracle
    rng = np.random.default_rng(9)
    for lmax, nmaps in [(300, 1), (257, 3)]:
        a = rng.standard_normal(((lmax + 1) ** 2, nmaps))
        assert rel(getSigmaL(a, lmax), cr_oracle.getSigmaL(a, lmax)) < 1e-12


def test_polarised_cr_path_gpu():
    """T,Q,U: 3 bands, CMB+synch with 3x3 S (TE != 0), Nside=64, lmax=128."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=64, lmax=128, comp_lmax=[128, 96], pol=True)
    S = oracle_system(spec)
    ctx = build_context(spec)
    x = np.random.default_rng(21).standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
    ctx.initPrecond(); ctx.update_precond(); S.init_precond_diag(); S.update_precond_diag()
    assert rel(ctx.cr_invM(x), S.invM(x)) < 1e-11
    resid, xi, eta = synth.draw_inputs(spec)
    rhso = S.computeRHS(resid, "sample", xi, eta)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), rhso) < 1e-11
    xs, n, st, res = ctx.solve_cr_eqn_by_CG(rhso, "fixed_iter", 1e-8, 5, 20, 1)
    xo, no, so = S.solve(rhso, "fixed_iter", 1e-8, 5, 20, 1)
    assert n == no == 20 and rel(xs, xo) < 1e-8


@pytest.mark.parametrize("nband", [9, 3, 6, 1, 2])
def test_spin2_adjoint_kernel_forms_gpu(nband, monkeypatch):
    """Polarised bands: three and more (Q,U) pairs per plan take the matrix-unit spin-2 adjoint (k_leg2_adj_mx, four
    pairs per launch: 9 = 4 + 4 + 1, 3 = one launch with an empty column group, 6 = 4 + 2), one or two (left over) pairs
    the VALU kernels of rounds 1-2 or, with CMDR_ADJ2_DX=1 / 2, the DPP form of the matrix-unit task (k_leg2_adj_dx) / its
    software-pipelined single-wave form with the sign-alternated recursion (k_leg2_adj_px) -- both correct but slower for
    one pair, off by default; CMDR_ADJ2_MX=0 sends everything to the VALU kernels.  All forms against the
    oracle, aniso noise (every (m, m') block of Yt N^-1 Y is populated), and against each other."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    cfg = dict(synth.CONFIGS["cfg3"])
    cfg["nu"], cfg["fwhm"] = cfg["nu"][:nband], cfg["fwhm"][:nband]
    spec = synth.make_problem(cfg, nside=128, lmax=200, pol=True, aniso=0.3)
    S = oracle_system(spec)
    ctx = build_context(spec)
    x = np.random.default_rng(100 + nband).standard_normal(ctx.ncr)
    want = S.matmulA(x)
    got_def = ctx.cr_matmulA(x)
    # synthesis: three and more pairs go four / three at a time through k_leg2_synth_npx; CMDR_SYNTH2_NP=0: the np2 kernels
    monkeypatch.setenv("CMDR_SYNTH2_NP", "0")
    got_s0 = ctx.cr_matmulA(x)
    monkeypatch.delenv("CMDR_SYNTH2_NP")
    assert np.array_equal(got_s0, got_def)     # the same operations in the same order per ring pair: bit-identical
    monkeypatch.setenv("CMDR_ADJ2_DX", "2")
    got_px = ctx.cr_matmulA(x)                                                         # matrix unit + pipelined DPP form
    assert rel(got_px, want) < 1e-11
    monkeypatch.setenv("CMDR_ADJ2_DX", "1")
    got_mx = ctx.cr_matmulA(x)                                                         # matrix unit + DPP form
    monkeypatch.setenv("CMDR_ADJ2_MX", "0")
    monkeypatch.setenv("CMDR_ADJ2_DX", "0")
    got_valu = ctx.cr_matmulA(x)
    assert rel(got_def, want) < 1e-11 and rel(got_mx, want) < 1e-11 and rel(got_valu, want) < 1e-11
    assert rel(got_mx, got_valu) < 1e-12 and not np.array_equal(got_mx, got_valu)     # different kernels did run
    if nband != 3:                                                                     # 3 pairs = one matrix-unit launch
        assert not np.array_equal(got_mx, got_def) and not np.array_equal(got_px, got_def)   # the DPP forms took the left-over pairs
    monkeypatch.delenv("CMDR_ADJ2_MX")
    monkeypatch.delenv("CMDR_ADJ2_DX")
    assert np.array_equal(ctx.cr_matmulA(x), got_def)                                  # deterministic


@pytest.mark.parametrize("cfg,kw", [("cfg2", dict(nside=64, lmax=128)), ("cfg2", dict(nside=32, lmax=64, pol=True)),
                                    ("cfg5", dict(nside=32, lmax=64, comp_lmax=[64, 48, 64, 40, 40]))])
def test_fixed_iter_graph_replay_equals_eager_launches_gpu(cfg, kw, monkeypatch):
    """fixed_iter solves replay iterations 3.. as one captured hipGraph of two PCG iterations (cr_system.cpp); the result
    must be the eager loop's bit for bit -- even and odd iteration counts, T / T,Q,U / varying mixing -- and the oracle's."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem(cfg, **kw)
    ctx = build_context(spec)
    ctx.initPrecond(); ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    for nit in (11, 12):
        monkeypatch.setenv("CMDR_CG_GRAPH", "0")
        x0, n0, _, res0 = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, nit, 1)
        monkeypatch.setenv("CMDR_CG_GRAPH", "1")
        x1, n1, _, res1 = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, nit, 1)
        assert n0 == n1 == nit and np.array_equal(x0, x1) and res0 == res1
    S = oracle_system(spec)
    S.init_precond_diag(); S.update_precond_diag()
    xo, no, _ = S.solve(b, "fixed_iter", 1e-8, 5, 12, 1)
    assert rel(x1, xo) < 1e-8


def test_varying_mixing_and_pseudoinv_gpu():
    """BASELINE.json configs[4] shape at reduced size (Nside=32, lmax=64): five diffuse components, synchrotron and
    dust with spatially varying spectral indices (Y . F . YtW branch of evalDiffuseBand / projectDiffuseBand,
    comm_diffuse_comp_mod.f90:2082-2084, 2155-2157), pseudo-inverse preconditioner (:2238-2380)."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg5", nside=32, lmax=64, comp_lmax=[64, 48, 64, 40, 40])
    S = oracle_system(spec)
    ctx = build_context(spec)
    x = np.random.default_rng(55).standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
    resid, xi, eta = synth.draw_inputs(spec)
    rhso = S.computeRHS(resid, "sample", xi, eta)
    assert rel(ctx.cr_computeRHS("sample", resid, xi, eta), rhso) < 1e-11
    ctx.initPrecond("pseudoinv"); ctx.update_precond()
    S.init_precond_pseudoinv(); S.update_precond_pseudoinv()
    for ib, b in enumerate(S.bands):
        assert np.allclose(ctx.alpha_nu(ib), b.alpha_nu, rtol=1e-11)
    assert rel(ctx.cr_invM(x), S.invM(x)) < 1e-10
    xs, n, st, res = ctx.solve_cr_eqn_by_CG(rhso, "fixed_iter", 1e-8, 5, 8, 1)
    xo, no, so = S.solve(rhso, "fixed_iter", 1e-8, 5, 8, 1)
    assert n == no == 8 and rel(xs, xo) < 1e-8


def test_varying_mixing_polarised_gpu():
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=32, lmax=64, pol=True)
    z = healpix.pix_z(32)
    spec["comps"][1]["F_map"] = {ib: np.repeat(synth.mixing("synch", b["nu"], z)[:, None], 3, axis=1) * [1.0, 1.02, 0.97]
                                 for ib, b in enumerate(spec["bands"])}
    S = oracle_system(spec)
    ctx = build_context(spec)
    x = np.random.default_rng(56).standard_normal(ctx.ncr)
    assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
    ctx.initPrecond("pseudoinv"); ctx.update_precond()
    S.init_precond_pseudoinv(); S.update_precond_pseudoinv()
    assert rel(ctx.cr_invM(x), S.invM(x)) < 1e-10


def test_edge_cases_gpu():
    from helpers import edge_case_checks
    edge_case_checks()


def test_chain_order_gpu():
    from commander_amd.cr import alm_to_chain_order, alm_from_chain_order
    from oracle import healpix
    rng = np.random.default_rng(12)
    for lmax, nmaps in [(0, 1), (33, 1), (300, 3)]:
        info = healpix.AlmInfo(lmax)
        a = rng.standard_normal((info.nalm, nmaps))
        c = alm_to_chain_order(a, lmax)
        ref = np.zeros_like(a, dtype=np.float32)
        ref[info.l ** 2 + info.l + info.m] = a.astype(np.float32)
        assert np.array_equal(c, ref)
        assert np.array_equal(alm_from_chain_order(c, lmax), a.astype(np.float32).astype(np.float64))


def test_cfg5_full_size_properties():
    """BASELINE.json configs[4] shape at full size (9 bands, Nside 1024, lmax 2000, five components, synchrotron and
    dust with spatially varying mixing, pseudo-inverse preconditioner): size-independent properties.  With unit ring
    weights YtW = (4pi/Npix) Yt, so even the varying-mixing operator is symmetric; A = 1 + positive semi-definite;
    a short pseudo-inverse-preconditioned PCG reduces the preconditioned residual monotonically."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg5")
    ctx = build_context(spec)
    rng = np.random.default_rng(17)
    x, y = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Ax, Ay = ctx.cr_matmulA(x), ctx.cr_matmulA(y)
    lhs, rhs = float(y @ Ax), float(x @ Ay)
    assert abs(lhs - rhs) <= 1e-10 * np.linalg.norm(y) * np.linalg.norm(Ax)
    assert float(x @ Ax) > float(x @ x) * (1 - 1e-12)
    ctx.initPrecond("pseudoinv")
    ctx.update_precond()
    Mx, My = ctx.cr_invM(x), ctx.cr_invM(y)
    assert abs(float(y @ Mx) - float(x @ My)) <= 1e-10 * np.linalg.norm(y) * np.linalg.norm(Mx)   # M^-1 symmetric
    assert float(x @ Mx) > 0.0
    res = []
    for nit in (1, 4):
        _, niter, stat, r = ctx.solve_cr_eqn_by_CG(Ax, conv_crit="fixed_iter", maxiter=nit)
        assert stat == 0 and niter == nit
        res.append(r[0] / r[1])
    assert res[1] < res[0] < 1.0


@pytest.mark.parametrize("pol", [False, True])
def test_compact_components_gpu(pol):
    """Templates (monopole + dipole per band) and point sources in the solve, Nside 32 / lmax 64, against the oracle."""
    import test_host_logic
    test_host_logic._compact_case(None, 32, 64, pol=pol)


def test_compact_components_cfg3_size():
    """The same at the benchmark geometry (9 bands, Nside 1024, lmax 2000): monopole + dipole templates on three of the
    bands (12 amplitudes) + 50 sources on all; the operator stays symmetric and >= 1, and a short PCG reduces the
    preconditioned residual."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.add_compact_blocks(synth.make_problem("cfg3"), nsrc=50, template_bands=(0, 4, 8))
    ctx = build_context(spec)
    assert ctx.ncr == synth.ncr_of(spec) == 2001 ** 2 + 12 + 50
    rng = np.random.default_rng(8)
    x, y = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Ax, Ay = ctx.cr_matmulA(x), ctx.cr_matmulA(y)
    assert abs(float(y @ Ax) - float(x @ Ay)) <= 1e-10 * np.linalg.norm(y) * np.linalg.norm(Ax)
    assert float(x @ Ax) > float(x @ x) * (1 - 1e-12)
    ctx.initPrecond(); ctx.update_precond()
    _, niter, stat, r = ctx.solve_cr_eqn_by_CG(Ax, conv_crit="fixed_iter", maxiter=6)
    assert stat == 0 and niter == 6 and r[0] < 0.1 * r[1]


def test_qucov_band_gpu():
    """White-noise T,Q,U band at Nside 32 plus a dense-QU-covariance band at Nside 8 (comm_N_QUcov) against the oracle."""
    import test_host_logic
    test_host_logic._qucov_case(None, nside_hi=32, lmax_hi=64, nside_lo=8, lmax_lo=16)


@pytest.mark.gpu
def test_gibbs_loop_updates_gpu():
    """Two Gibbs iterations (amplitudes | C_l, C_l | amplitudes) through getSigmaL / sampleCls_binned / updateS /
    cmdr_comp_set_cl / update_precond, then the sampling-group and mixing updates, against the oracle."""
    from helpers import gibbs_loop_checks
    # converged solves of two fp64 implementations agree to the stated 1e-6, not to rounding: the CG
    # trajectories separate at the 1e-16 level and the criterion stops both at 1e-12 of the preconditioned residual
    gibbs_loop_checks(None, nside=16, lmax=32, tol=1e-6)


@pytest.mark.gpu
def test_context_lifecycle_releases_device_memory_gpu():
    """A chain creates and destroys solver contexts (new sampling groups, new band sets): device memory must come back.
    20 create / solve / destroy cycles at Nside 64 with compact blocks and a mixing map; free memory afterwards within
    16 MiB of the level after the first cycle (allocator granularity), i.e. nothing accumulates per cycle."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    from commander_amd.lib import device_mem_info
    spec = synth.make_problem("cfg2", nside=64, lmax=128, pol=True)
    synth.add_compact_blocks(spec, nsrc=3)
    levels = []
    for it in range(20):
        ctx = build_context(spec)
        ctx.initPrecond(); ctx.update_precond()
        resid, xi, eta = synth.draw_inputs(spec)
        b = ctx.cr_computeRHS("sample", resid, xi, eta)
        x, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=3)
        assert np.all(np.isfinite(x))
        ctx.close()
        levels.append(device_mem_info()[0])
    assert abs(levels[-1] - levels[1]) < 16 << 20, [(v - levels[1]) >> 20 for v in levels]


@pytest.mark.gpu
def test_compute_residual_gpu():
    """compute_residual(cg_samp_group) on the GPU (Nside 16, lmax 32; T and T,Q,U; constant and varying mixing, compact
    blocks outside the group) against the oracle."""
    from helpers import residual_checks
    residual_checks(None, nside=16, lmax=32)


@pytest.mark.gpu
def test_chisq_convergence_criterion_gpu():
    from helpers import chisq_criterion_checks
    chisq_criterion_checks(None, nside=16, lmax=32)


@pytest.mark.gpu
def test_compute_residual_and_chisq_full_size_properties_gpu():
    """BASELINE geometry (9 bands, Nside 1024, lmax 2000), five components, sampling group = CMB: properties that need
    no oracle.  compute_residual is affine in the amplitudes (linearity of the subtracted signal), ignores the group's own
    component; the 'chisq' criterion runs at this size without disturbing the iteration."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg5")
    for c in spec["comps"][1:]:
        c["active"] = False
    ctx = build_context(spec)
    rng = np.random.default_rng(8)
    na = (spec["comps"][0]["lmax"] + 1) ** 2
    a1, a2 = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    data = [rng.standard_normal(len(b["siN"])) for b in spec["bands"]]
    zero = [np.zeros(len(b["siN"])) for b in spec["bands"]]
    s1 = ctx.compute_residual(a1, zero)
    s2 = ctx.compute_residual(a2, zero)
    s12 = ctx.compute_residual(a1 + 2.0 * a2, data)
    for b in range(len(data)):
        want = data[b][:, None] + s1[b] + 2.0 * s2[b]
        assert rel(s12[b], want) < 1e-12
        assert np.abs(s1[b]).max() > 0
    a3 = a1.copy()
    a3[:na] = rng.standard_normal(na)                     # the group's own component does not enter
    s3 = ctx.compute_residual(a3, zero)
    for b in range(len(data)):
        assert np.array_equal(s3[b], s1[b])
    # the 'chisq' criterion at full size: evaluated at every check, cannot stop before miniter, reports non-convergence
    ctx.initPrecond(); ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    rhs = ctx.cr_computeRHS("sample", resid, xi, eta)
    x, n, stat, _ = ctx.solve_cr_eqn_by_CG(rhs, "chisq", 1e-30, 5, 3, 1)
    assert n == 3 and stat == 1 and np.all(np.isfinite(x))
    xf, nf, _, _ = ctx.solve_cr_eqn_by_CG(rhs, "fixed_iter", 1e-30, 5, 3, 1)
    assert nf == 3 and np.array_equal(x, xf)              # evaluating chisq does not disturb the iteration


def test_lowl_preconditioner_gpu():
    """CG_LMAX_PRECOND low-l dense block (updateLowlPrecond / applyLowlPrecond) on the GPU vs the oracle."""
    from helpers import lowl_precond_checks
    lowl_precond_checks(None, nside=32, lmax=64, L=8, nside_low=8)


def test_mono_dipole_prior_gpu():
    """applyMonoDipolePrior (comm_diffuse_comp_mod.f90:5738-5827) on the GPU vs the oracle."""
    from helpers import mono_dipole_prior_checks
    mono_dipole_prior_checks(None)


def test_literal_quirks_switch_gpu():
    from helpers import literal_quirks_checks
    literal_quirks_checks(None, nside=32, lmax=64)


def test_coefficients_formed_in_the_synthesis_staging_gpu():
    from helpers import fused_staging_checks
    fused_staging_checks(None, tol=1e-11)


@pytest.mark.parametrize("cfg,pol", [("cfg2", False), ("cfg2", True), ("cfg3", False)])
def test_fused_pcg_updates_equal_the_general_sequence_gpu(cfg, pol, monkeypatch):
    from helpers import fused_pcg_checks
    fused_pcg_checks(None, pol, monkeypatch, nside=32, lmax=64, cfg=cfg)


@pytest.mark.parametrize("env", [{}, {"CMDR_SYNTH_DPP": "0"}, {"CMDR_SYNTH_PREP": "0"},
                                 {"CMDR_SYNTH_DPP": "0", "CMDR_SYNTH_PREP": "0"}, {"CMDR_ADJ_X9": "0"}, {"CMDR_ADJ_DX": "0"},
                                 {"CMDR_SYNTH_M4": "3"}, {"CMDR_SYNTH_M4": "3", "CMDR_SYNTH_PREP": "0"},
                                 {"CMDR_SYNTH_M4": "3", "CMDR_UNIFORM_START": "0"}])
def test_kernel_form_switches_agree_with_the_oracle(env, monkeypatch):
    """The A/B switches of the Legendre kernels (synthesis through LDS broadcasts instead of DPP, coefficients from the
    stream instead of the staging, ninth map in its own launch, small batches through the VALU adjoint, synthesis on the
    small matrix-unit tile) select code paths that stay compiled in: each must give the oracle's matvec, on 9 bands
    (matrix unit) and on 4 (DPP / VALU form)."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for nband in (9, 4):
        spec = synth.make_problem("cfg3", nside=128, lmax=200, bands=list(range(nband)))
        ctx = build_context(spec)
        S = oracle_system(spec)
        x = np.random.default_rng(3 + nband).standard_normal(ctx.ncr)
        assert rel(ctx.cr_matmulA(x), S.matmulA(x)) < 1e-11
        ctx.close()


def test_page_locked_caller_buffers_give_the_same_rhs_gpu():
    """cmdr_host_register / cmdr_host_unregister: the host-pointer entry points read page-locked caller arrays (DMA) and
    pageable ones (staged) to the same bits."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg3", nside=64, lmax=128)
    ctx = build_context(spec)
    ctx.initPrecond()
    ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    want = ctx.cr_computeRHS("sample", resid, xi, eta)
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in list(resid) + list(xi) + [eta]]
    for a in arrs:
        ctx.host_register(a)
    try:
        nb = len(resid)
        got = ctx.cr_computeRHS("sample", arrs[:nb], arrs[nb:2 * nb], arrs[-1])
        sol = ctx.solve_cr_eqn_by_CG(ctx.host_register(got), "fixed_iter", maxiter=5)[0]
        ctx.host_unregister(got)
    finally:
        for a in arrs:
            ctx.host_unregister(a)
    assert np.array_equal(got, want)
    assert np.array_equal(sol, ctx.solve_cr_eqn_by_CG(want, "fixed_iter", maxiter=5)[0])
    ctx.close()
