"""Generates the committed golden vectors of tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

Nothing upstream pins this path (the reference has no tests and no fixtures: SURVEY.md §4), so the goldens are
produced by checkers that are independent of both the oracle's recursions/FFTs and the HIP kernels:
  * sht_bruteforce_nside{4,8}.npz : dense direct sums over scipy.special.sph_harm_y (oracle/bruteforce.py)
  * invn_diag_3j.npz              : compute_invN_lm evaluated literally with exact Racah-formula 3j symbols
  * invn_diag_3j_map.npz          : the same together with the noise map it belongs to (pins the product's pipeline)
  * lm2i_tables.json              : Commander's a_lm index maps for lmax=4, P=1 and P=3 (comm_map_mod.f90:228-261)
  * kat.json                      : the reference's own 2x2 PCG known-answer test and the fiducial dipole constants
  * mini_commander.json           : what fortran/mini_commander.f90 must print: the CPU oracle (oracle/cr_oracle.py)
                                    run on the same problem and the same LCG draws (python tests/golden/make_golden.py mini)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import bruteforce, healpix, sht, wigner  # noqa: E402


def main():
    rng = np.random.default_rng(20261004)
    for nside, lmax in [(4, 8), (8, 20)]:
        B = bruteforce.basis_matrix(nside, lmax)
        w = 1.0 + 0.05 * rng.standard_normal(2 * nside)
        wp = bruteforce.ring_weight_per_pixel(nside, w)
        a = rng.standard_normal((lmax + 1) ** 2)
        m = rng.standard_normal(12 * nside * nside)
        np.savez_compressed(os.path.join(HERE, "sht_bruteforce_nside%d.npz" % nside), nside=nside, lmax=lmax, wring=w,
                            alm=a, map=m, Y=B @ a, Yt=B.T @ m, YtW=B.T @ (wp * m), WY=wp * (B @ a))
    # spin-2 (Q,U <-> E,B): dense matrix from the Goldberg closed form of +-2Y_lm (no recursion)
    nside, lmax = 4, 8
    B2 = bruteforce.basis_matrix_spin2(nside, lmax)
    info = healpix.AlmInfo(lmax)
    e, b = rng.standard_normal(info.nalm), rng.standard_normal(info.nalm)
    e[info.l < 2] = 0.0
    b[info.l < 2] = 0.0
    mq, mu = rng.standard_normal(12 * nside * nside), rng.standard_normal(12 * nside * nside)
    qu = B2 @ np.concatenate([e, b])
    eb = B2.T @ np.concatenate([mq, mu])
    np.savez_compressed(os.path.join(HERE, "sht_spin2_bruteforce_nside4.npz"), nside=nside, lmax=lmax, almE=e, almB=b,
                        mapQ=mq, mapU=mu, Y_Q=qu[: mq.size], Y_U=qu[mq.size:], Yt_E=eb[: e.size], Yt_B=eb[e.size:])
    nside, lmax = 4, 12
    siN2 = 1.0 + 0.5 * rng.random(12 * nside * nside)
    al0 = sht.YtW(nside, lmax, siN2)[: lmax + 1]
    np.savez_compressed(os.path.join(HERE, "invn_diag_3j.npz"), nside=nside, lmax=lmax, al0=al0,
                        diag=wigner.invn_diag_3j(nside, lmax, al0))
    tables = {}
    for P in (1, 3):
        for r in range(P):
            info = healpix.AlmInfo(4, r, P)
            tables["P%d_r%d" % (P, r)] = {"lm": info.lm.T.tolist(), "mind": info.mind.tolist(), "nalm": info.nalm}
    json.dump(tables, open(os.path.join(HERE, "lm2i_tables.json"), "w"), indent=1)
    kat = {
        # commander3/todscripts/wmap/cg_solver.py:54-61
        "pcg2x2": {"A": [[3, 2], [2, 6]], "b": [2, -8], "x": [2, -2]},
        # commander3/src/comm_chisq_mod.f90:296-301 (uK, (l,m) = (1,-1),(1,0),(1,1))
        "fiducial_dipole_uK": [-4.54107e3, 5.119744e3, 4.848587e2],
    }
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1)


def invn_map():
    """invn_diag_3j_map.npz: a noise map siN^2 together with compute_invN_lm (comm_N_mod.f90:127-197) evaluated
    literally with exact 3j symbols from the map's a_l0 -- pins the product's whole invN_diag pipeline (YtW of the map,
    then the diagonal), not only the oracle's quadrature.  Own seed, so the older fixtures stay byte-identical."""
    rng = np.random.default_rng(20261005)
    nside, lmax = 4, 12
    siN2 = 1.0 + 0.5 * rng.random(12 * nside * nside)
    al0 = sht.YtW(nside, lmax, siN2)[: lmax + 1]
    np.savez_compressed(os.path.join(HERE, "invn_diag_3j_map.npz"), nside=nside, lmax=lmax, siN2=siN2, al0=al0,
                        diag=wigner.invn_diag_3j(nside, lmax, al0))


def mini_commander():
    """mini_commander.json: the third amplitude sample of fortran/mini_commander.f90 (Nside 64, lmax 128, one band, CMB T;
    cr_computeRHS + 50 fixed PCG iterations + applyMonoDipolePrior 'monopole+dipole') computed by the oracle from the
    same inputs: the driver's minimal-standard LCG + Box-Muller draws in the driver's order."""
    from oracle import cr_oracle as cro
    nside, lmax = 64, 128
    npix, nalm = 12 * nside * nside, (lmax + 1) ** 2
    l = np.arange(lmax + 1)
    sigma = (60.0 / 60.0 * np.pi / 180.0) / np.sqrt(8.0 * np.log(2.0))
    b_l = np.exp(-0.5 * l * (l + 1.0) * sigma ** 2)
    Dl = np.full((lmax + 1, 1), 1000.0)
    z = 1.0 - 2.0 * (np.arange(1, npix + 1) - 0.5) / npix
    siN = 1.0 / (40.0 * (1.0 + 0.5 * z))
    siN[np.abs(z) < 0.2] = 0.0
    cl = cro.Cl(lmax, 1, Dl)
    # the driver's own S tables: C_0 = D_0, C_l = D_l 2 pi / (l (l + 1))
    Cl_drv = np.where(l == 0, 1000.0, 1000.0 * 2.0 * np.pi / np.maximum(l * (l + 1.0), 1.0))
    assert np.allclose(cl.sqrtS_mat[0, 0, :], np.sqrt(Cl_drv), rtol=1e-14)
    S = cro.CRSystem([cro.Band(nside, lmax, siN[:, None], b_l[:, None])], [cro.DiffuseComp(lmax, 1, cl, [[1.0]])])
    S.init_precond_diag()
    state = [163425]

    def uni():
        state[0] = (16807 * state[0]) % 2147483647
        return state[0] / 2147483647.0

    def gauss():
        u1 = uni(); u2 = uni()
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    mask = (siN > 0.0).astype(np.float64)
    for it in range(3):
        resid, xi = np.zeros(npix), np.zeros(npix)
        for i in range(npix):
            if siN[i] > 0.0:
                resid[i] = gauss() / siN[i]
            xi[i] = gauss()
        eta = np.array([gauss() for _ in range(nalm)])
        if it < 2:
            continue                       # the samples are independent; only the draws of the earlier ones matter
        rhs = S.computeRHS([resid[:, None]], "sample", [xi[:, None]], eta)
        S.update_precond_diag()
        x, niter, _ = S.solve(rhs, "fixed_iter", 1e-8, 5, 50, 1)
        amp, mu = S.apply_mono_dipole_prior(0, x, nside, mask, "monopole+dipole", b_l)
    out = {"nside": nside, "lmax": lmax, "niter": int(niter), "mu": [float(v) for v in mu],
           "norm": float(np.linalg.norm(amp)), "amp_first8": [float(v) for v in amp[:8]],
           "generator": "python tests/golden/make_golden.py mini  (oracle/cr_oracle.py)"}
    json.dump(out, open(os.path.join(HERE, "mini_commander.json"), "w"), indent=1)
    print(out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "invn_map":
        invn_map()
    elif len(sys.argv) > 1 and sys.argv[1] == "mini":
        mini_commander()
    else:
        main()
        invn_map()
