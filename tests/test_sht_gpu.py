"""GPU parity: HIP SHT (through the C ABI) vs the CPU oracle on identical seeded inputs.
Tolerances: single SHT rel-L2 <= 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-11


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("nside,lmax", [(4, 8), (4, 11), (8, 23), (16, 32), (64, 128), (64, 191), (256, 512)])
def test_sht_all_jobs_vs_oracle(nside, lmax, oracle_lib):
    from commander_amd import ShtPlan
    rng = np.random.default_rng(1000 + nside + lmax)
    w = 1.0 + 0.05 * rng.standard_normal(2 * nside)
    plan = ShtPlan(nside, lmax, wring=w, max_maps=2)
    a = rng.standard_normal(((lmax + 1) ** 2, 2))
    m = rng.standard_normal((12 * nside * nside, 2))
    y = plan.Y(a)
    yt = plan.Yt(m)
    ytw = plan.YtW(m)
    wy = plan.WY(a)
    for k in range(2):
        assert rel(y[:, k], oracle_lib.Y(nside, lmax, a[:, k])) < TOL
        assert rel(yt[:, k], oracle_lib.Yt(nside, lmax, m[:, k])) < TOL
        assert rel(ytw[:, k], oracle_lib.YtW(nside, lmax, m[:, k], wring=w)) < TOL
        assert rel(wy[:, k], oracle_lib.WY(nside, lmax, a[:, k], wring=w)) < TOL


def test_sht_adjointness_full_size():
    """Size-independent property at the benchmark geometry: <Y a, m> == <a, Yt m>."""
    from commander_amd import ShtPlan
    nside, lmax = 1024, 2000
    rng = np.random.default_rng(7)
    plan = ShtPlan(nside, lmax)
    a = rng.standard_normal((lmax + 1) ** 2)
    m = rng.standard_normal(12 * nside * nside)
    lhs = plan.Y(a) @ m
    rhs = a @ plan.Yt(m)
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))


def test_sht_ring_subset_matches_full(oracle_lib):
    from commander_amd import ShtPlan
    from oracle import healpix
    nside, lmax, P = 32, 64, 3
    rng = np.random.default_rng(5)
    a = rng.standard_normal((lmax + 1) ** 2)
    ref = oracle_lib.Y(nside, lmax, a)
    acc = np.zeros((lmax + 1) ** 2)
    mfull = rng.standard_normal(12 * nside * nside)
    for r in range(P):
        rings = np.arange(1 + r, 2 * nside + 1, P, dtype=np.int32)
        allr = sorted(list(rings) + [4 * nside - i for i in rings if i < 2 * nside])
        idx = np.concatenate([np.arange(healpix.ring_info(nside, i)[4], healpix.ring_info(nside, i)[4]
                                        + healpix.ring_info(nside, i)[0]) for i in allr])
        plan = ShtPlan(nside, lmax, rings=rings)
        assert plan.npix == idx.size
        assert rel(plan.Y(a), ref[idx]) < TOL
        acc += plan.Yt(mfull[idx])
    assert rel(acc, oracle_lib.Yt(nside, lmax, mfull)) < TOL


@pytest.mark.parametrize("nside,lmax", [(4, 8), (16, 47), (64, 128), (256, 512)])
def test_sht_spin2_vs_oracle(nside, lmax, oracle_lib):
    """(Q,U) <-> (E,B): Commander's polarisation call (comm_map_mod.f90:446-449)."""
    from commander_amd import ShtPlan
    rng = np.random.default_rng(77 + nside)
    w = 1.0 + 0.05 * rng.standard_normal(2 * nside)
    plan = ShtPlan(nside, lmax, wring=w, max_maps=2, pol=True)
    na, npx = (lmax + 1) ** 2, 12 * nside * nside
    e, b = rng.standard_normal(na), rng.standard_normal(na)
    mq, mu = rng.standard_normal(npx), rng.standard_normal(npx)
    q, u = plan.execute_spin2(1, almE=e, almB=b)
    assert rel(np.concatenate([q, u]), np.concatenate(oracle_lib.sht_spin2(1, nside, lmax, almE=e, almB=b))) < TOL
    ee, bb = plan.execute_spin2(2, mapQ=mq, mapU=mu)
    assert rel(np.concatenate([ee, bb]), np.concatenate(oracle_lib.sht_spin2(2, nside, lmax, mapQ=mq, mapU=mu))) < TOL
    ee, bb = plan.execute_spin2(0, mapQ=mq, mapU=mu)
    assert rel(np.concatenate([ee, bb]), np.concatenate(oracle_lib.sht_spin2(0, nside, lmax, mapQ=mq, mapU=mu, wring=w))) < TOL
    q, u = plan.execute_spin2(3, almE=e, almB=b)
    assert rel(np.concatenate([q, u]), np.concatenate(oracle_lib.sht_spin2(3, nside, lmax, almE=e, almB=b, wring=w))) < TOL


def test_sht_spin2_adjointness_full_size():
    from commander_amd import ShtPlan
    nside, lmax = 1024, 2000
    rng = np.random.default_rng(8)
    plan = ShtPlan(nside, lmax, max_maps=2, pol=True)
    na, npx = (lmax + 1) ** 2, 12 * nside * nside
    e, b = rng.standard_normal(na), rng.standard_normal(na)
    e[:2] = 0.0; b[:2] = 0.0   # (l < 2, m = 0)
    e[lmax + 1:lmax + 3] = 0.0; b[lmax + 1:lmax + 3] = 0.0   # (l = 1, m = +-1)
    mq, mu = rng.standard_normal(npx), rng.standard_normal(npx)
    q, u = plan.execute_spin2(1, almE=e, almB=b)
    ee, bb = plan.execute_spin2(2, mapQ=mq, mapU=mu)
    lhs, rhs = q @ mq + u @ mu, e @ ee + b @ bb
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))


def test_gpu_vs_golden_vectors():
    """HIP path against the committed brute-force golden vectors directly (no oracle in between)."""
    from helpers import golden_checks, golden_kat
    golden_checks()
    golden_kat()


@pytest.mark.parametrize("R", [4, 2])
def test_matrix_unit_adjoint_both_task_sizes(R, oracle_lib, monkeypatch):
    """9 maps per call: 8 go through the matrix-unit adjoint (256-pair tasks at 4 ring pairs per lane, 128-pair tasks
    at 2 -- what ring-sharded ranks with few pairs get), the ninth through the VALU kernel; both against the oracle,
    on a ring subset (every 2nd ring pair: the per-64-pair skip of not-yet-started groups sees mixed latitudes)."""
    from commander_amd.sht import ShtPlan
    monkeypatch.setenv("CMDR_LEG_R", str(R))
    nside, lmax = 256, 400
    rng = np.random.default_rng(R)
    plan = ShtPlan(nside, lmax, max_maps=9)
    m = rng.standard_normal((12 * nside * nside, 9))
    yt = plan.Yt(m)
    for k in (0, 5, 7, 8):
        assert rel(yt[:, k], oracle_lib.Yt(nside, lmax, m[:, k])) < 1e-12
    rings = np.arange(1, 2 * nside + 1, 2, dtype=np.int32)
    sub = ShtPlan(nside, lmax, rings=rings, max_maps=9)
    from commander_amd import healpix
    pix = healpix.local_pixels(nside, rings)
    full = np.zeros((12 * nside * nside, 9))
    full[pix] = m[pix]
    yts = sub.Yt(np.ascontiguousarray(m[pix]))
    for k in (0, 8):
        assert rel(yts[:, k], oracle_lib.Yt(nside, lmax, full[:, k])) < 1e-12


@pytest.mark.parametrize("R", [4, 2])
def test_dpp_adjoint_all_batch_sizes(R, oracle_lib, monkeypatch):
    """3..5 maps per call go through k_leg_adj_dx (the matrix-unit kernel's task with every map accumulated by DPP
    row-broadcast FMAs; 6 and 7 too once CMDR_ADJ_MX is raised), at 4 and 2 ring pairs per lane; 1, 2 and the 2 left over
    of 10 through the VALU kernel."""
    from commander_amd.sht import ShtPlan
    monkeypatch.setenv("CMDR_LEG_R", str(R))
    monkeypatch.setenv("CMDR_ADJ_MX", "8")
    nside, lmax = 256, 300
    rng = np.random.default_rng(40 + R)
    m = rng.standard_normal((12 * nside * nside, 10))
    ref = {k: oracle_lib.Yt(nside, lmax, m[:, k]) for k in (0, 3, 6, 9)}
    for nm in (1, 2, 3, 4, 5, 6, 7, 10):
        plan = ShtPlan(nside, lmax, max_maps=nm)
        yt = plan.Yt(np.ascontiguousarray(m[:, :nm]))
        if nm == 1:
            yt = yt.reshape(-1, 1)
        for k in ref:
            if k < nm:
                assert rel(yt[:, k], ref[k]) < 1e-12, (nm, k)
        plan.close()


@pytest.mark.parametrize("uniform", ["1", "0"])
def test_block_starts_uniform_and_per_lane(uniform, oracle_lib, monkeypatch):
    """The plan lets the 64 ring pairs of a lane block switch on at one l == m (mod 32) with their true, still tiny mu as
    seeds (the kernels then inject seeds once per block); CMDR_UNIFORM_START=0 keeps the per-lane starts and the kernels'
    per-l start tests.  Both against the oracle, synthesis and adjoint, 9 maps (matrix unit + ninth map) and 4 (DPP form)."""
    from commander_amd.sht import ShtPlan
    monkeypatch.setenv("CMDR_UNIFORM_START", uniform)
    nside, lmax = 256, 500
    rng = np.random.default_rng(77)
    for nm in (9, 4):
        plan = ShtPlan(nside, lmax, max_maps=nm)
        a = rng.standard_normal(((lmax + 1) ** 2, nm))
        m = rng.standard_normal((12 * nside * nside, nm))
        y, yt = plan.Y(a), plan.Yt(m)
        for k in (0, nm - 1):
            assert rel(y[:, k], oracle_lib.Y(nside, lmax, a[:, k])) < 1e-12
            assert rel(yt[:, k], oracle_lib.Yt(nside, lmax, m[:, k])) < 1e-12
        plan.close()
