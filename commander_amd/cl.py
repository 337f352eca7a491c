"""Host-side mirror of ``comm_Cl%updateS`` (commander3/src/comm_Cl_mod.f90:316-384): per-l nmaps x nmaps
S_mat / sqrtS_mat / sqrtInvS_mat from D_l.  Setup-time CPU work in the reference as well (LAPACK dsyevd through
``compute_hermitian_root``, commander3/src/math_tools.f90:606-662); the GPU only consumes the tables.

Used by ``synth`` to build synthetic problems.  The library's own entry point for a Fortran caller is
``cmdr_cl_update_S`` (``commander_amd.cr.updateS``), checked against the oracle in tests/test_cl_step.py."""
import numpy as np


def _hermitian_root(A, pw):
    W, V = np.linalg.eigh(A)
    if np.any(W <= 0):
        out = A.copy()
        out[0, 0] = -1e30      # math_tools.f90:640-648
        return out
    return (V * W ** pw) @ V.T


def update_S(Dl, nmaps, lmin=0, RJ2unit=None):
    """Dl: (lmax+1, nspec) with nspec = nmaps(nmaps+1)/2 in Commander's (TT,TE,TB,EE,EB,BB) order.
    Returns (sqrtS_mat, sqrtInvS_mat, S_mat), each (nmaps, nmaps, lmax+1) Fortran-ordered."""
    Dl = np.asarray(Dl, dtype=np.float64)
    lmax = Dl.shape[0] - 1
    Dl = Dl.reshape(lmax + 1, -1)
    RJ = np.ones(nmaps) if RJ2unit is None else np.asarray(RJ2unit, dtype=np.float64)
    sq = np.zeros((nmaps, nmaps, lmax + 1), order="F")
    isq = np.zeros((nmaps, nmaps, lmax + 1), order="F")
    S = np.zeros((nmaps, nmaps, lmax + 1), order="F")
    for l in range(lmax + 1):
        M = np.zeros((nmaps, nmaps))
        ok = np.zeros(nmaps, dtype=bool)
        k = 0
        for i in range(nmaps):
            for j in range(i, nmaps):
                if l < lmin:
                    v = 0.0
                elif l == 0:
                    v = Dl[l, k]
                else:
                    v = Dl[l, k] / (l * (l + 1) / (2.0 * np.pi))
                M[i, j] = M[j, i] = v / (RJ[i] * RJ[j])
                if i == j:
                    ok[i] = Dl[l, k] > 0.0
                k += 1
        for i in range(nmaps):
            if not ok[i]:
                M[i, :] = 0.0
                M[:, i] = 0.0
                M[i, i] = 1.0
        a, b = _hermitian_root(M, 0.5), _hermitian_root(M, -0.5)
        for i in range(nmaps):
            if not ok[i]:
                a[i, :] = a[:, i] = 0.0
                b[i, :] = b[:, i] = 0.0
        sq[:, :, l], isq[:, :, l], S[:, :, l] = a, b, a @ a
    return sq, isq, S
