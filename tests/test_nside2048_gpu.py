"""GPU tier: BASELINE.json configs[3] geometry (Nside=2048, lmax=4000; polarised).  The long cap rings
(4096 < nphi < 8192, not a power of two) do not fit one LDS image and run as two half-length Bluestein transforms
(kernels_body.hpp ring_block, split branch); this file checks that path at full size against the oracle and through
size-independent properties (exact adjointness, analysis(synthesis) round trip)."""
import numpy as np
import pytest

from helpers import rel

pytestmark = pytest.mark.gpu

NSIDE, LMAX = 2048, 4000


@pytest.fixture(scope="module")
def plan():
    from commander_amd.sht import ShtPlan
    return ShtPlan(NSIDE, LMAX, max_maps=2, pol=True)


def test_spin0_full_size_vs_oracle(plan, oracle_lib):
    rng = np.random.default_rng(2048)
    a = rng.standard_normal(((LMAX + 1) ** 2, 1))
    m = rng.standard_normal((12 * NSIDE * NSIDE, 1))
    y, yt = plan.Y(a), plan.Yt(m)
    assert rel(y[:, 0], oracle_lib.Y(NSIDE, LMAX, a[:, 0])) < 1e-11
    assert rel(yt[:, 0], oracle_lib.Yt(NSIDE, LMAX, m[:, 0])) < 1e-11
    # exact adjointness <Yt m, a> == <m, Y a>
    lhs, rhs = float(yt[:, 0] @ a[:, 0]), float(m[:, 0] @ y[:, 0])
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs), np.linalg.norm(m) * np.linalg.norm(y))
    # band-limited round trip: YtW Y a == a to quadrature accuracy (HEALPix, unit ring weights: ~1e-3 at lmax ~ 2 nside)
    back = plan.YtW(y)
    assert rel(back[:, 0], a[:, 0]) < 2e-2


def test_spin2_full_size_vs_oracle(plan, oracle_lib):
    from oracle import sht as osht
    rng = np.random.default_rng(4000)
    na = (LMAX + 1) ** 2
    e, b = rng.standard_normal(na), rng.standard_normal(na)
    for v in (e, b):   # l < 2 carries no spin-2 signal: m=0 block l=0,1 and the m=1 pair at l=1
        v[0:2] = 0.0
        v[LMAX + 1:LMAX + 3] = 0.0
    q, u = plan.execute_spin2(osht.JOB_Y, almE=e, almB=b)
    qo, uo = osht.sht_spin2(osht.JOB_Y, NSIDE, LMAX, almE=e, almB=b)
    assert rel(q, qo) < 1e-11 and rel(u, uo) < 1e-11
    mq, mu = rng.standard_normal(q.size), rng.standard_normal(q.size)
    et, bt = plan.execute_spin2(osht.JOB_Yt, mapQ=mq, mapU=mu)
    lhs = float(et @ e + bt @ b)
    rhs = float(mq @ q + mu @ u)
    assert abs(lhs - rhs) <= 1e-10 * np.sqrt((mq @ mq + mu @ mu) * (q @ q + u @ u))


def test_cfg4_polarised_cr_full_size_properties():
    """BASELINE.json configs[3] (polarised T/E/B CMB-only, Nside 2048, lmax 4000) through the CR-level ABI: A is
    symmetric (size-independent property of cr_matmulA, which uses Yt not YtW for exactly that reason,
    comm_cr_mod.f90:771-1024), and a short fixed_iter PCG reduces the preconditioned residual."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg4")
    ctx = build_context(spec)
    assert ctx.ncr == 3 * (LMAX + 1) ** 2
    rng = np.random.default_rng(7)
    x, y = rng.standard_normal(ctx.ncr), rng.standard_normal(ctx.ncr)
    Ax, Ay = ctx.cr_matmulA(x), ctx.cr_matmulA(y)
    lhs, rhs = float(y @ Ax), float(x @ Ay)
    assert abs(lhs - rhs) <= 1e-11 * np.linalg.norm(y) * np.linalg.norm(Ax)
    assert float(x @ Ax) > float(x @ x) * (1 - 1e-12)      # A = 1 + (positive semi-definite)
    ctx.initPrecond()
    ctx.update_precond()
    xs, niter, stat, res = ctx.solve_cr_eqn_by_CG(Ax, conv_crit="fixed_iter", maxiter=8)
    assert stat == 0 and niter == 8
    assert res[0] < 1e-2 * res[1]
