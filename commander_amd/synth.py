"""Synthetic workloads for the CR solver (no real Planck data is available offline).

Every value here is this build's own fixed choice so all sessions / ranks use identical inputs:
  * CMB prior   D_l^TT = 1000 uK^2 flat (l >= 1), D_0 = D_1         (power_law type, comm_Cl_mod.f90:226-246)
  * synchrotron D_l = 100 (l/80)^-2.5, constant beta = -3.1, nu_ref = 30 GHz  =>  F_mean = (nu/30)^-3.1
  * Gaussian beams b_l = exp(-l(l+1) sigma^2/2)  (gaussbeam path, comm_utils.f90:91-92), no pixel window
  * noise rms_p = sigma_0 (1 + 0.5 cos theta_p), sigma_0 from S/N = 1 at the beam scale; mask |cos theta| < sin 12 deg
  * random numbers: counter-based Philox, master seed 163425 (tutorial/param_tutorial.txt:15), one sub-stream per
    (kind, band|component), drawn in global RING / stacked order so results do not depend on the GPU count.
"""
import numpy as np

from . import cl as _cl
from . import healpix

BASE_SEED = 163425

PLANCK_NU = [30.0, 44.0, 70.0, 100.0, 143.0, 217.0, 353.0, 545.0, 857.0]
PLANCK_FWHM = [32.3, 27.0, 13.2, 9.7, 7.3, 5.0, 4.9, 4.8, 4.6]

CONFIGS = {
    # BASELINE.json configs[0..2] (spin-0 / temperature)
    "cfg1": dict(nside=64, lmax=128, nu=[70.0], fwhm=[60.0], comps=["cmb"]),
    "cfg2": dict(nside=256, lmax=512, nu=[30.0, 70.0, 143.0], fwhm=[32.3, 13.2, 7.3], comps=["cmb", "synch"]),
    "cfg3": dict(nside=1024, lmax=2000, nu=PLANCK_NU, fwhm=PLANCK_FWHM, comps=["cmb"]),
    # configs[3]: polarised T/E/B CMB-only at the SHT-roofline geometry
    "cfg4": dict(nside=2048, lmax=4000, nu=[143.0], fwhm=[7.3], comps=["cmb"], pol=True),
    # configs[4]: BeyondPlanck-like full diffuse model; synchrotron and dust have spatially varying spectral indices
    # (Y . F . YtW mixing branch), free-free and AME constant ones; pseudo-inverse preconditioner territory
    "cfg5": dict(nside=1024, lmax=2000, nu=PLANCK_NU, fwhm=PLANCK_FWHM, comps=["cmb", "synch", "dust", "ff", "ame"],
                 varying=["synch", "dust"]),
}

H_OVER_K = 0.0479924466  # K / GHz


def mixing(name, nu, z):
    """Synthetic mixing matrix F_nu(p) of a component (relative to its reference frequency); z = cos(theta) of the
    pixels, or None for the spatially constant version (F_mean)."""
    zz = 0.0 if z is None else z
    if name == "cmb":
        return 1.0 + 0.0 * zz
    if name == "synch":
        return (nu / 30.0) ** (-3.1 + 0.1 * zz)
    if name == "dust":   # modified blackbody, beta_d = 1.55 + 0.05 cos(theta), T_d = 19.6 K, nu_ref = 545 GHz
        x, x0 = H_OVER_K * nu / 19.6, H_OVER_K * 545.0 / 19.6
        return (nu / 545.0) ** (1.55 + 0.05 * zz + 1.0) * (np.exp(x0) - 1.0) / (np.exp(x) - 1.0)
    if name == "ff":
        return (nu / 40.0) ** -2.14 + 0.0 * zz
    if name == "ame":
        return np.exp(-0.5 * (np.log(nu / 22.0) / 0.6) ** 2) + 0.0 * zz
    raise ValueError(name)


def gaussbeam(fwhm_arcmin, lmax):
    sigma = np.radians(fwhm_arcmin / 60.0) / np.sqrt(8.0 * np.log(2.0))
    l = np.arange(lmax + 1, dtype=np.float64)
    return np.exp(-0.5 * l * (l + 1.0) * sigma * sigma)


def rng(kind, index):
    return np.random.Generator(np.random.Philox(key=BASE_SEED, counter=[0, 0, int(kind), int(index)]))


def comp_Dl(name, lmax):
    l = np.arange(lmax + 1, dtype=np.float64)
    if name == "cmb":
        D = np.full(lmax + 1, 1000.0)
    elif name == "synch":
        D = 100.0 * (np.maximum(l, 1.0) / 80.0) ** -2.5
    elif name == "dust":
        D = 300.0 * (np.maximum(l, 1.0) / 80.0) ** -2.6
    elif name == "ff":
        D = 30.0 * (np.maximum(l, 1.0) / 80.0) ** -2.2
    elif name == "ame":
        D = 30.0 * (np.maximum(l, 1.0) / 80.0) ** -2.4
    else:
        raise ValueError(name)
    D[0] = D[1] if lmax >= 1 else D[0]
    return D


def noise_rms(nside, sigma0, aniso=0.0):
    """rms map of a band: sigma_0 (1 + 0.5 cos theta), optionally modulated in azimuth
    (aniso > 0: x (1 + aniso cos(3 phi + 1) sin theta), the test variant whose N^-1 couples different m)."""
    z = healpix.pix_z(nside)
    rms = sigma0 * (1.0 + 0.5 * z)
    if aniso:
        rms = rms * (1.0 + aniso * np.cos(3.0 * healpix.pix_phi(nside) + 1.0) * np.sqrt(1.0 - z * z))
    return rms


def noise_mask(nside, aniso=0.0):
    """True where siN = 0: |cos theta| < sin 12 deg; aniso > 0 tilts the band in azimuth (a
    galactic-plane-like cut |z - 0.3 sin(phi)| < sin 12 deg) and adds a few round holes off the plane."""
    z = healpix.pix_z(nside)
    if not aniso:
        return np.abs(z) < np.sin(np.radians(12.0))
    phi = healpix.pix_phi(nside)
    m = np.abs(z - 0.3 * np.sin(phi)) < np.sin(np.radians(12.0))
    sth = np.sqrt(1.0 - z * z)
    for z0, p0, rad in ((0.7, 0.5, 6.0), (-0.55, 2.5, 4.0), (0.95, 4.0, 5.0), (-0.98, 1.0, 3.0)):
        s0 = np.sqrt(1.0 - z0 * z0)
        m |= (sth * s0 * np.cos(phi - p0) + z * z0) > np.cos(np.radians(rad))
    return m


def make_problem(cfg, nside=None, lmax=None, comp_lmax=None, pixels=None, pol=None, bands=None, aniso=0.0):
    """Problem spec dict consumed by ``commander_amd.cr.build_context`` (and by the tests' oracle builder).

    aniso: 0 = the benchmark noise of SURVEY.md 8d (rms and mask functions of cos theta only); > 0 = the parity-test
    variant with azimuth-dependent rms and mask (every (m, m') block of Yt N^-1 Y is then populated).
    pixels: optional full-sky RING indices of a rank's local map (ring sharding); maps are then local.
    bands:  optional subset of band indices this rank holds (band sharding); F_mean / F_map rows follow."""
    band_subset = bands
    c = dict(CONFIGS[cfg]) if isinstance(cfg, str) else dict(cfg)
    nside = int(nside or c["nside"])
    lmax = int(lmax or c["lmax"])
    pol = bool(c.get("pol", False)) if pol is None else bool(pol)
    npix = 12 * nside * nside
    z = healpix.pix_z(nside)
    Dl_cmb = comp_Dl("cmb", lmax)
    l = np.arange(lmax + 1, dtype=np.float64)
    Cl_cmb = np.where(l > 0, Dl_cmb * 2.0 * np.pi / np.maximum(l * (l + 1.0), 1.0), Dl_cmb)
    bands = []
    for nu, fwhm in zip(c["nu"], c["fwhm"]):
        b_l = gaussbeam(fwhm, lmax)
        below = np.nonzero(b_l ** 2 < 0.5)[0]
        l_half = int(below[0]) if below.size else lmax
        l_star = max(1, min(l_half, int(0.75 * lmax)))
        sigma0 = np.sqrt(Cl_cmb[l_star] * b_l[l_star] ** 2 * npix / (4.0 * np.pi))
        rms = noise_rms(nside, sigma0, aniso)
        siN = 1.0 / rms
        siN[noise_mask(nside, aniso)] = 0.0
        if pixels is not None:
            siN = siN[pixels]
        if pol:   # T, Q, U: polarisation noise sqrt(2) higher, same beam (read_beam default, comm_utils.f90:103-107)
            siN = np.stack([siN, siN / np.sqrt(2.0), siN / np.sqrt(2.0)], axis=1)
            b_l = np.stack([b_l, b_l, b_l], axis=1)
        bands.append(dict(nside=nside, lmax=lmax, nu=nu, fwhm=fwhm, siN=siN, b_l=b_l, mb_eff=1.0, sigma0=sigma0))
    comps = []
    for k, name in enumerate(c["comps"]):
        cl_lmax = lmax if comp_lmax is None else int(comp_lmax[k])
        nm = 3 if pol else 1
        Dl = comp_Dl(name, cl_lmax)[:, None]
        if pol:   # (TT, TE, TB, EE, EB, BB): EE = 1e-3 TT, BB = 1e-4 TT for l >= 2, a small TE to exercise the 3x3 roots
            lv = np.arange(cl_lmax + 1)
            pm = (lv >= 2).astype(np.float64)          # Dl(0:1, 2:) = 0 (comm_Cl_mod.f90:217)
            tt = Dl[:, 0]
            Dl = np.stack([tt, 0.01 * tt * pm, 0.0 * tt, 1e-3 * tt * pm, 0.0 * tt, 1e-4 * tt * pm], axis=1)
        sq, isq, S = _cl.update_S(Dl, nm)
        F = np.repeat(np.array([float(mixing(name, b["nu"], None)) for b in bands])[:, None], nm, axis=1)
        comp = dict(name=name, lmax=cl_lmax, nmaps=nm, F_mean=F, sqrtS_mat=sq, sqrtInvS_mat=isq, S_mat=S,
                    Dl=Dl if pol else Dl[:, 0], active=True)
        if name in c.get("varying", []):
            zl = z if pixels is None else z[pixels]
            comp["F_map"] = {ib: np.repeat(mixing(name, b["nu"], zl)[:, None], nm, axis=1) for ib, b in enumerate(bands)}
            # F_mean = full-sky pixel average of the map (comm_diffuse_comp_mod.f90:1991-1999)
            comp["F_mean"] = np.repeat(np.array([mixing(name, b["nu"], z).mean() for b in bands])[:, None], nm, axis=1)
        comps.append(comp)
    band_ids = list(range(len(bands)))
    if band_subset is not None:
        band_ids = [int(b) for b in band_subset]
        bands = [bands[b] for b in band_ids]
        for comp in comps:
            comp["F_mean"] = comp["F_mean"][band_ids, :]
            if "F_map" in comp:
                comp["F_map"] = {i: comp["F_map"][b] for i, b in enumerate(band_ids)}
    return dict(bands=bands, comps=comps, nside=nside, lmax=lmax, pixels=pixels, band_ids=band_ids, aniso=aniso)


def ncr_of(spec):
    return sum(c["nparam"] if c.get("kind") == "compact" else (c["lmax"] + 1) ** 2 * c["nmaps"] for c in spec["comps"])


def add_compact_blocks(spec, nsrc=5, seed=77, template_bands=None):
    """Append synthetic compact components to a problem spec: a monopole + dipole template block
    on every band (comm_md_comp-like: 4 amplitudes per band, dense columns) inserted after the first diffuse component,
    and ``nsrc`` point sources (sparse beam-sized footprints on every band, flat spectrum) at the end."""
    import scipy.sparse as sp
    nside = spec["nside"]
    npix_full = 12 * nside * nside
    pix = spec["pixels"] if spec["pixels"] is not None else np.arange(npix_full)
    # unit vectors of the local pixels (RING): z from pix_z, phi from the ring geometry
    z = healpix.pix_z(nside)
    phi = np.concatenate([healpix.ring_info(nside, r)[2] + 2.0 * np.pi * np.arange(healpix.ring_info(nside, r)[0])
                          / healpix.ring_info(nside, r)[0] for r in range(1, 4 * nside)])
    sth = np.sqrt(1.0 - z * z)
    vec = np.stack([np.ones(npix_full), sth * np.cos(phi), sth * np.sin(phi), z], axis=1)[pix]
    np_loc = pix.size
    blocks = []
    for ib, b in enumerate(spec["bands"]):
        if template_bands is not None and ib not in template_bands:
            continue
        nm = b["siN"].shape[1] if np.ndim(b["siN"]) > 1 else 1
        T = np.zeros((np_loc * nm, 4))
        T[:np_loc, :] = vec                          # temperature-only templates (Stokes 0 cells)
        blocks.append(dict(kind="compact", nparam=4, sigma=[30.0, 10.0, 10.0, 10.0], mean=[1.0, 0.0, 0.5, -0.5],
                           P={ib: sp.csr_matrix(T)}))
    g = np.random.Generator(np.random.Philox(key=BASE_SEED, counter=[0, 0, 9, seed]))
    centres = g.choice(npix_full, nsrc, replace=False)
    P = {}
    for ib, b in enumerate(spec["bands"]):
        nm = b["siN"].shape[1] if np.ndim(b["siN"]) > 1 else 1
        rows, cols, vals = [], [], []
        for s_, c0 in enumerate(centres):
            d2 = 2.0 - 2.0 * (vec_full_dot(z, phi, c0))        # squared chord distance to the source
            sig = max(np.radians(b["fwhm"] / 60.0) / np.sqrt(8.0 * np.log(2.0)), 1.5 * np.sqrt(4.0 * np.pi / npix_full))
            near = np.nonzero(d2[pix] < (3.0 * sig) ** 2)[0]
            rows += list(near)
            cols += [s_] * near.size
            vals += list(np.exp(-0.5 * d2[pix][near] / sig ** 2))
        P[ib] = sp.csr_matrix((vals, (rows, cols)), shape=(np_loc * nm, nsrc))
    src = dict(kind="compact", nparam=nsrc, sigma=5.0, mean=0.0, P=P)
    spec["comps"] = spec["comps"][:1] + blocks + spec["comps"][1:] + [src]
    return spec


def vec_full_dot(z, phi, p0):
    sth, s0 = np.sqrt(1.0 - z * z), np.sqrt(1.0 - z[p0] ** 2)
    return sth * s0 * np.cos(phi - phi[p0]) + z * z[p0]


def draw_inputs(spec):
    """Residual maps d_nu = rms * xi_d (noise-only data; signal content does not change the operator or the cost),
    RHS noise draws xi and prior draws eta, all unit Gaussians from fixed Philox sub-streams."""
    nside = spec["nside"]
    npix = 12 * nside * nside
    pix = spec["pixels"]
    resid, xi = [], []
    ids = spec.get("band_ids") or list(range(len(spec["bands"])))
    for b, i in zip(spec["bands"], ids):          # sub-streams follow the GLOBAL band index (band sharding)
        g = rng(1, i).standard_normal(npix)
        rms = noise_rms(nside, b["sigma0"], spec.get("aniso", 0.0))
        d = rms * g
        x = rng(2, i).standard_normal(npix)
        nm = np.ndim(b["siN"]) > 1 and b["siN"].shape[1] or 1
        if nm > 1:
            d = np.stack([d] + [np.sqrt(2.0) * rms * rng(4 + j, i).standard_normal(npix) for j in range(1, nm)], axis=1)
            x = np.stack([x] + [rng(6 + j, i).standard_normal(npix) for j in range(1, nm)], axis=1)
        if pix is not None:
            d, x = d[pix], x[pix]
        resid.append(d)
        xi.append(x)
    eta = rng(3, 0).standard_normal(ncr_of(spec))
    return resid, xi, eta


def lowres_noise(spec, nside_lowres):
    """``data(b)%N%siN_lowres`` of every band (comm_N_rms_mod.f90:250-259): sqrt(udgrade(siN^2)) * nside / nside_lowres,
    i.e. the square root of the summed inverse variance of a coarse pixel's children.  Needs full-sky bands (the driver
    side of a sharded run builds it before scattering the maps)."""
    out = []
    for b in spec["bands"]:
        s = np.asarray(b["siN"], dtype=np.float64).reshape(12 * b["nside"] ** 2, -1)[:, 0]
        ns = min(int(nside_lowres), b["nside"])
        out.append((ns, np.sqrt(healpix.udgrade_sum_ring(s * s, b["nside"], ns))))
    return out
