// Element bodies of the harmonic-space / vector kernels of the CR solver (host-emulable, see kernels_body.hpp).
// Each works on one (l, m) of an (l, m) grid and touches both the (+m) and (-m) slots of Commander's real-packed
// layout (commander3/src/comm_map_mod.f90:228-261).
#pragma once
#include "kernels_body.hpp"

namespace cmdr {

struct CompDev {        // one diffuse component inside the stacked vector (comm_cr_utils.f90:25-33 ind_comp)
    long long pos;      // 0-based start in x
    long long nalm;     // (lmax+1)^2
    int lmax;           // lmax_amp
    int nmaps;
    int lmax_cl;        // lmax of the S tables (-1: cltype 'none')
    int active;
    long long smat_off; // offset of this component's [3 kinds][nmaps*nmaps*(lmax_cl+1)] tables (sqrtS, sqrtInvS, S)
};

// out = f(M_l) applied per l to the nmaps-vector of a component (comm_Cl_mod.f90:588-674), optionally + add.
//   kind 0: sqrtS_mat, 1: sqrtInvS_mat.  l > lmax_cl -> 0.  cltype 'none' (lmax_cl < 0) -> identity.
//   inactive components: out = (add ? add : 0)  [cr_matmulA never touches their slots: comm_cr_mod.f90:800-803]
CMDR_HD void sqrtS_slot(const CompDev& C, const double* __restrict__ smat, int kind, const double* __restrict__ in,
                        const double* __restrict__ add, double* __restrict__ out, int l, int64_t i,
                        bool pass_inactive) {   // i = packed index of one real slot of multipole l
    const int nm = C.nmaps;
    double v[3] = {0.0, 0.0, 0.0}, r[3] = {0.0, 0.0, 0.0};
    for (int a = 0; a < nm; ++a) v[a] = in[C.pos + a * C.nalm + i];
    if (!C.active) {
        for (int a = 0; a < nm; ++a) r[a] = pass_inactive ? v[a] : 0.0;
    } else if (C.lmax_cl < 0) {
        for (int a = 0; a < nm; ++a) r[a] = v[a];
    } else if (l <= C.lmax_cl) {
        const double* M = smat + C.smat_off + (int64_t)kind * nm * nm * (C.lmax_cl + 1) + (int64_t)nm * nm * l;
        for (int a = 0; a < nm; ++a) {
            double s = 0.0;
            for (int b = 0; b < nm; ++b) s += M[a + nm * b] * v[b];
            r[a] = s;
        }
    }
    for (int a = 0; a < nm; ++a) {
        double o = r[a];
        // the unit prior / eta / mu terms exist only for active components with a prior
        // (comm_cr_mod.f90:698, :967: "if (trim(c%cltype) /= 'none')")
        if (add) o += (C.active && C.lmax_cl >= 0) ? add[C.pos + a * C.nalm + i] : 0.0;
        out[C.pos + a * C.nalm + i] = o;
    }
}
CMDR_HD void sqrtS_elem(const CompDev& C, const double* __restrict__ smat, int kind, const double* __restrict__ in,
                        const double* __restrict__ add, double* __restrict__ out, int m, int l,
                        bool pass_inactive) {
    const int64_t i0 = d_packed_index(C.lmax, l, m);
    sqrtS_slot(C, smat, kind, in, add, out, l, i0, pass_inactive);
    if (m > 0) sqrtS_slot(C, smat, kind, in, add, out, l, i0 + 1, pass_inactive);
}

// Band stream entry: a~_bm(l,m) = cnorm * kappa_m * sum_c w[bm][c][l] * sx_{c, stokes(bm)}(l,m)
//   w folds F_mean * b_l * mb_eff and both l-truncations (comm_cr_mod.f90:858-867,
//   comm_diffuse_comp_mod.f90:2077-2089, comm_B_bl_mod.f90:108-127).
// One real part (part 0 = re, 1 = im) of the coefficient-stream entry (l, m) of band map bm: the beam- and mixing-
// weighted sum over the components, times the recursion normalisation.  Shared by k_band_prep (stream written to
// memory) and the synthesis kernel's tile staging (stream never written: PrepDev).
CMDR_HD double band_prep_part(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ sx,
                              const double* __restrict__ w /* [ncomp][lmax_g+1] for this bm */, int stokes,
                              const double* __restrict__ cnorm, int lmax_g, int m, int l, int part,
                              const double* __restrict__ extra /* packed a_lm(lmax_g) added as is, or null */) {
    if (l > lmax_g || (part && m == 0)) return 0.0;
    double v = 0.0;
    if (extra) v = extra[d_packed_index(lmax_g, l, m) + part];   // components with spatially varying mixing
    for (int c = 0; c < ncomp; ++c) {
        const CompDev C = comps[c];
        if (l > C.lmax || stokes >= C.nmaps) continue;
        const double wc = w[(int64_t)c * (lmax_g + 1) + l];
        if (wc == 0.0) continue;
        v += wc * sx[C.pos + (int64_t)stokes * C.nalm + d_packed_index(C.lmax, l, m) + part];
    }
    return v * (cnorm[d_moffp(lmax_g, m) + (l - m)] * (m == 0 ? 1.0 : 0.70710678118654752440));
}
CMDR_HD void band_prep_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ sx,
                            const double* __restrict__ w /* [ncomp][lmax_g+1] for this bm */, int stokes,
                            double* __restrict__ ast_base, int nbs, int bm, const double* __restrict__ cnorm,
                            int lmax_g, int m, int l,
                            const double* __restrict__ extra = nullptr /* packed a_lm(lmax_g) added as is */) {
    const int64_t t = d_moffp(lmax_g, m) + (l - m);
    double* __restrict__ ast = ast_base + 2 * (t * nbs + bm) - 2 * t;  // maps interleaved: slot of (t, bm)
    ast[2 * t] = band_prep_part(comps, ncomp, sx, w, stokes, cnorm, lmax_g, m, l, 0, extra);
    ast[2 * t + 1] = band_prep_part(comps, ncomp, sx, w, stokes, cnorm, lmax_g, m, l, 1, extra);
}

// Where the synthesis finds its coefficients when the stream is not materialised (launch_leg_synth, prep != null)
struct PrepDev {
    const CompDev* comps;
    int ncomp;
    const double* sx;        // S^1/2 x, stacked vector
    const double* w;         // [nbm][ncomp][lmax+1]
    const int* bm_stokes;    // [nbm]
    const double* cnorm;     // padded-triangle normalisation of the plan
    const double* extra;     // [nbm][(lmax+1)^2] or null
};

// Component entry of y_c: (+)= kappa'_m * sum_{bm in group} w[bm][c][l] cnorm[t] sum_chunks part[bm][chunk][t]
//   (projectDiffuseBand, comm_diffuse_comp_mod.f90:2112-2167, and the truncation comm_cr_mod.f90:931-933).
CMDR_HD void band_post_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ part,
                            int64_t part_map_stride, int64_t part_chunk_stride, int nchunk, int nbm,
                            const int* __restrict__ bm_stokes, const double* __restrict__ w /* [nbm][ncomp][lmax_g+1] */,
                            const double* __restrict__ cnorm, int lmax_g, double* __restrict__ yc, int accumulate,
                            int m, int l, const int* __restrict__ lwtab = nullptr) {
    // one thread = one (l, m): the chunk partials of a band map are read once and feed every component
    constexpr int kMaxComp = 8;
    double re[kMaxComp], im[kMaxComp];
    for (int c = 0; c < kMaxComp; ++c) re[c] = im[c] = 0.0;
    if (l <= lmax_g) {
        const int64_t t = d_moffp(lmax_g, m) + (l - m);
        for (int b = 0; b < nbm; ++b) {
            if (bm_stokes[b] != 0) continue;
            double wc[kMaxComp];
            bool any = false;
            for (int c = 0; c < ncomp; ++c) {
                wc[c] = comps[c].active ? w[((int64_t)b * ncomp + c) * (lmax_g + 1) + l] : 0.0;
                any = any || wc[c] != 0.0;
            }
            if (!any) continue;
            const double* p = part + b * part_map_stride + 2 * t;
            double sr = 0.0, si = 0.0;
            for (int ch = 0; ch < nchunk; ++ch) {
                if (lwtab && l < lwtab[m * nchunk + ch]) continue;   // never written by the adjoint: structurally zero
                sr += p[ch * part_chunk_stride];
                si += p[ch * part_chunk_stride + 1];
            }
            for (int c = 0; c < ncomp; ++c)
                if (wc[c] != 0.0) { re[c] += wc[c] * sr; im[c] += wc[c] * si; }
        }
        const double f = cnorm[t] * (m == 0 ? 1.0 : 1.41421356237309504880);
        for (int c = 0; c < ncomp; ++c) { re[c] *= f; im[c] *= f; }
    }
    for (int c = 0; c < ncomp; ++c) {
        const CompDev C = comps[c];
        if (m > C.lmax || l > C.lmax) continue;
        const int64_t i0 = d_packed_index(C.lmax, l, m);
        for (int a = 0; a < C.nmaps; ++a) {
            const double vr = a == 0 ? re[c] : 0.0, vi = a == 0 ? im[c] : 0.0;   // (Q,U) columns come from band_post2
            const int64_t i = C.pos + (int64_t)a * C.nalm + i0;
            if (accumulate) {
                yc[i] += vr;
                if (m > 0) yc[i + 1] += vi;
            } else {
                yc[i] = vr;
                if (m > 0) yc[i + 1] = vi;
            }
        }
    }
}

// Polarisation pair ip of a band: (E, B) stream entry from the components' Stokes columns 2 and 3 (0-based 1, 2):
//   E' = -cnorm2 kappa_m / 2 * sum_c wE[c][l] sx_{c,E}(l,m),  B' likewise (comm_map_mod.f90:446-449 spin-2 call on 2:3)
CMDR_HD void band_prep2_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ sx,
                             const double* __restrict__ wE, const double* __restrict__ wB, double* __restrict__ st,
                             int npol, int ip, const double* __restrict__ cnorm2, int lmax_g, int m, int l,
                             const double* __restrict__ extraE = nullptr, const double* __restrict__ extraB = nullptr) {
    const int64_t t = d_moffp(lmax_g, m) + (l - m);
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (l <= lmax_g && l >= 2) {
        if (extraE) {
            const int64_t i = d_packed_index(lmax_g, l, m);
            v[0] = extraE[i];
            v[2] = extraB[i];
            if (m > 0) { v[1] = extraE[i + 1]; v[3] = extraB[i + 1]; }
        }
        for (int c = 0; c < ncomp; ++c) {
            const CompDev C = comps[c];
            if (l > C.lmax || C.nmaps < 3) continue;
            const int64_t i = d_packed_index(C.lmax, l, m);
            const double we = wE[(int64_t)c * (lmax_g + 1) + l], wb = wB[(int64_t)c * (lmax_g + 1) + l];
            const double* e = sx + C.pos + 1 * C.nalm + i;
            const double* b = sx + C.pos + 2 * C.nalm + i;
            v[0] += we * e[0];
            v[2] += wb * b[0];
            if (m > 0) { v[1] += we * e[1]; v[3] += wb * b[1]; }
        }
        const double f = -0.5 * cnorm2[t] * (m == 0 ? 1.0 : 0.70710678118654752440);
        for (int k = 0; k < 4; ++k) v[k] *= f;
    }
    double* o = st + 4 * (t * npol + ip);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}

// Transpose of band_prep2_elem: component c, Stokes E and B, accumulated into yc (always +=)
CMDR_HD void band_post2_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ part2,
                             int64_t part_pol_stride, int64_t part_chunk_stride, int nchunk, int npol,
                             const double* __restrict__ w /* [nbm][ncomp][lmax_g+1] */, int nT,
                             const double* __restrict__ cnorm2, int lmax_g, double* __restrict__ yc, int m, int l,
                             const int* __restrict__ lwtab = nullptr) {
    // one thread = one (l, m): the chunk partials of a polarisation pair are read once and feed every component
    if (l > lmax_g || l < 2) return;
    constexpr int kMaxComp = 8;
    const int64_t t = d_moffp(lmax_g, m) + (l - m);
    double s[kMaxComp][4];
    bool use[kMaxComp];
    bool any = false;
    for (int c = 0; c < ncomp; ++c) {
        const CompDev& C = comps[c];
        use[c] = C.nmaps == 3 && C.active && m <= C.lmax && l <= C.lmax;
        any = any || use[c];
        for (int k = 0; k < 4; ++k) s[c][k] = 0.0;
    }
    if (!any) return;
    for (int ip = 0; ip < npol; ++ip) {
        const double* p = part2 + ip * part_pol_stride + 4 * t;
        double a[4] = {0.0, 0.0, 0.0, 0.0};
        for (int ch = 0; ch < nchunk; ++ch) {
            if (lwtab && l < lwtab[m * nchunk + ch]) continue;     // never written: structurally zero
            for (int k = 0; k < 4; ++k) a[k] += p[ch * part_chunk_stride + k];
        }
        for (int c = 0; c < ncomp; ++c) {
            if (!use[c]) continue;
            const double we = w[((int64_t)(nT + 2 * ip) * ncomp + c) * (lmax_g + 1) + l];
            const double wb = w[((int64_t)(nT + 2 * ip + 1) * ncomp + c) * (lmax_g + 1) + l];
            s[c][0] += we * a[0]; s[c][1] += we * a[1]; s[c][2] += wb * a[2]; s[c][3] += wb * a[3];
        }
    }
    const double f = -0.5 * cnorm2[t] * (m == 0 ? 1.0 : 1.41421356237309504880);
    for (int c = 0; c < ncomp; ++c) {
        if (!use[c]) continue;
        const CompDev& C = comps[c];
        const int64_t i = d_packed_index(C.lmax, l, m);
        double* e = yc + C.pos + 1 * C.nalm + i;
        double* b = yc + C.pos + 2 * C.nalm + i;
        e[0] += s[c][0] * f;
        b[0] += s[c][2] * f;
        if (m > 0) { e[1] += s[c][1] * f; b[1] += s[c][3] * f; }
    }
}

// alm_equal with an optional per-l factor (comm_map_mod.f90:1148-1165 + comm_B_bl_mod.f90:108-127):
//   dst(l,m) (+)= f_l * src(l,m) for l <= min(lmax_s, lmax_d, lcut); without accumulate the rest of dst is zero-filled.
CMDR_HD void alm_copy_elem(const double* __restrict__ src, int lmax_s, double* __restrict__ dst, int lmax_d,
                           const double* __restrict__ fl, int accumulate, int lcut, int m, int l) {
    const int64_t id = d_packed_index(lmax_d, l, m);
    double re = 0.0, im = 0.0;
    if (l <= lmax_s && l <= lcut) {
        const int64_t is = d_packed_index(lmax_s, l, m);
        const double f = fl ? fl[l] : 1.0;
        re = f * src[is];
        if (m > 0) im = f * src[is + 1];
    }
    if (accumulate) {
        dst[id] += re;
        if (m > 0) dst[id + 1] += im;
    } else {
        dst[id] = re;
        if (m > 0) dst[id + 1] = im;
    }
}

// One column of a batched a_lm copy (launch_alm_copy_batch): the arguments of alm_copy_elem.  The columns of one
// launch must write distinct destinations.
struct AlmCopyDesc {
    const double* src;
    double* dst;
    const double* fl;
    int lmax_s, lmax_d, accumulate, lcut;
};
constexpr int kAlmCopyBatch = 16;
struct AlmCopyBatch {
    AlmCopyDesc d[kAlmCopyBatch];
};

// Prior part and scatter of the pseudo-inverse preconditioner (applyDiffPrecond_pseudoinv,
// comm_diffuse_comp_mod.f90:2328-2372): out_c = z_c + sum_c' Q[stokes][c][c'][l] x_c' for active components
// (Q = B B^t, B = prior columns of pinv(U)); inactive components keep x.
CMDR_HD void pinv_prior_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ Q, int lmax_pre,
                             int nmaps_pre, const double* __restrict__ x, const double* __restrict__ z,
                             double* __restrict__ out, int m, int l) {
    const int nslot = m == 0 ? 1 : 2;
    for (int j = 0; j < nmaps_pre; ++j)
        for (int sl = 0; sl < nslot; ++sl) {
            double v[8];
            for (int k = 0; k < ncomp; ++k) {
                const CompDev C = comps[k];
                v[k] = (C.active && l <= C.lmax && j < C.nmaps)
                           ? x[C.pos + (int64_t)j * C.nalm + d_packed_index(C.lmax, l, m) + sl] : 0.0;
            }
            for (int k1 = 0; k1 < ncomp; ++k1) {
                const CompDev C = comps[k1];
                if (l > C.lmax || j >= C.nmaps) continue;
                const int64_t i = C.pos + (int64_t)j * C.nalm + d_packed_index(C.lmax, l, m) + sl;
                if (!C.active) { out[i] = x[i]; continue; }
                double s = z[i];
                for (int k2 = 0; k2 < ncomp; ++k2)
                    s += Q[(((int64_t)j * ncomp + k1) * ncomp + k2) * (lmax_pre + 1) + l] * v[k2];
                out[i] = s;
            }
        }
}

// Compact components (templates, point sources; comm_template_comp_mod.f90:210-270, comm_ptsrc_comp_mod.f90:336-428):
// parameter p adds val * a_p to the cells listed for it.  cell = pix + npix * stokes of ONE band; the band's Stokes
// maps live at base[stokes] inside a plan's map buffer (T maps first, then the (Q,U) pairs).
struct CellBase {
    int64_t np;         // pixels per map
    int64_t off[3];     // map offsets (in doubles) of the band's T, Q, U maps inside the buffer
};
CMDR_HD int64_t cell_addr(const CellBase& B, int64_t cell) {
    const int64_t st = cell / B.np;
    return B.off[st] + (cell - st * B.np);
}
// CSR row (= cell): z[cell] += sum_k val[k] a[col[k]]
CMDR_HD void compact_fwd_row(double* __restrict__ z, const CellBase& B, const int64_t* __restrict__ rows,
                             const int64_t* __restrict__ ptr, const int* __restrict__ col,
                             const double* __restrict__ val, const double* __restrict__ a, int64_t r) {
    double s = 0.0;
    for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) s += val[k] * a[col[k]];
    z[cell_addr(B, rows[r])] += s;
}
// one CSC entry of column p: val * u[cell]
CMDR_HD double compact_adj_term(const double* __restrict__ u, const CellBase& B, const int64_t* __restrict__ cell,
                                const double* __restrict__ val, int64_t k) {
    return val[k] * u[cell_addr(B, cell[k])];
}

// Chain-file order of a_lm (comm_map_mod.f90:712-719: ind = l^2 + l + m, m = -l..l, single precision) <-> packed.
//   to_chain != 0: out32[l^2+l+m] = (float) alm[packed(l, m)]   ;   else: alm[packed(l, m)] = in32[l^2+l+m]
CMDR_HD void alm_chain_elem(double* __restrict__ alm, float* __restrict__ c32, int lmax, int to_chain, int m, int l) {
    const int64_t i = d_packed_index(lmax, l, m);
    const int64_t ip = (int64_t)l * l + l + m, im = (int64_t)l * l + l - m;
    if (to_chain) {
        c32[ip] = (float)alm[i];
        if (m > 0) c32[im] = (float)alm[i + 1];
    } else {
        alm[i] = (double)c32[ip];
        if (m > 0) alm[i + 1] = (double)c32[im];
    }
}

// Diagonal preconditioner (applyDiffPrecond_diagonal, comm_diffuse_comp_mod.f90:2186-2235): per (l, m, stokes)
// a dense npre x npre block (same for +m and -m).  P layout: [stokes][k1][k2][ntri(lmax_pre)] (unpadded triangle).
CMDR_HD int64_t d_moff(int lmax, int m) { return (int64_t)m * (lmax + 1) - (int64_t)m * (m - 1) / 2; }

CMDR_HD void precond_diag_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ P,
                               int lmax_pre, int nmaps_pre, const double* __restrict__ in, double* __restrict__ out,
                               int m, int l) {
    const int64_t ntri = (int64_t)(lmax_pre + 1) * (lmax_pre + 2) / 2;
    const int64_t t = d_moff(lmax_pre, m) + (l - m);
    const int nslot = m == 0 ? 1 : 2;
    for (int j = 0; j < nmaps_pre; ++j) {
        for (int sl = 0; sl < nslot; ++sl) {
            double v[8];
            for (int k = 0; k < ncomp; ++k) {
                const CompDev C = comps[k];
                v[k] = (l <= C.lmax && j < C.nmaps) ? in[C.pos + (int64_t)j * C.nalm + d_packed_index(C.lmax, l, m) + sl]
                                                     : 0.0;
            }
            for (int k1 = 0; k1 < ncomp; ++k1) {
                const CompDev C = comps[k1];
                if (l > C.lmax || j >= C.nmaps) continue;
                double s = 0.0;
                for (int k2 = 0; k2 < ncomp; ++k2)
                    s += P[(((int64_t)j * ncomp + k1) * ncomp + k2) * ntri + t] * v[k2];
                out[C.pos + (int64_t)j * C.nalm + d_packed_index(C.lmax, l, m) + sl] = s;
            }
        }
    }
}

// ---- the PCG vector updates fused per (l, m) entry (solve_cr_eqn_by_CG, comm_cr_mod.f90:253-272); each routine handles
// every slot of one (l, m) -- all components, Stokes blocks, re and im -- which is the granularity at which S^1/2 and
// the diagonal preconditioner couple the stacked vector.  Used by k_cg_q / k_cg_xr / k_cg_d and the host emulation.
template <class F>
CMDR_HD void cg_for_slots(const CompDev* __restrict__ comps, int ncomp, int m, int l, F f) {
    for (int c = 0; c < ncomp; ++c) {
        const CompDev C = comps[c];
        if (l > C.lmax) continue;
        const int64_t i0 = C.pos + d_packed_index(C.lmax, l, m);
        for (int a = 0; a < C.nmaps; ++a) {
            f(i0 + a * C.nalm);
            if (m > 0) f(i0 + a * C.nalm + 1);
        }
    }
}
// q = S^1/2 yc + d (the tail of cr_matmulA, :957-1008); returns this entry's share of d.q (:254)
CMDR_HD double cg_q_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ smat,
                         const double* __restrict__ yc, const double* __restrict__ d, double* __restrict__ q, int m,
                         int l) {
    for (int c = 0; c < ncomp; ++c)
        if (l <= comps[c].lmax) sqrtS_elem(comps[c], smat, 0, yc, d, q, m, l, false);
    double acc = 0.0;
    cg_for_slots(comps, ncomp, m, l, [&](int64_t i) { acc += d[i] * q[i]; });
    return acc;
}
// x += alpha d ; r -= alpha q (:255-261) ; s = M^-1 r, diagonal type (:266) ; returns the share of r.s (:269)
CMDR_HD double cg_xr_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ P, int lmax_pre,
                          int nmaps_pre, double alpha, double* __restrict__ x, double* r,
                          const double* __restrict__ d, const double* __restrict__ q, double* s, int m, int l) {
    cg_for_slots(comps, ncomp, m, l, [&](int64_t i) {
        x[i] += alpha * d[i];
        r[i] -= alpha * q[i];
    });
    precond_diag_elem(comps, ncomp, P, lmax_pre, nmaps_pre, r, s, m, l);
    double acc = 0.0;
    cg_for_slots(comps, ncomp, m, l, [&](int64_t i) { acc += r[i] * s[i]; });
    return acc;
}
// d = s + beta d (:271-272) ; sx = S^1/2 d (the head of the next cr_matmulA, :792-836)
CMDR_HD void cg_d_elem(const CompDev* __restrict__ comps, int ncomp, const double* __restrict__ smat, double beta,
                       double* d, const double* __restrict__ s, double* __restrict__ sx, int m, int l) {
    cg_for_slots(comps, ncomp, m, l, [&](int64_t i) { d[i] = s[i] + beta * d[i]; });
    for (int c = 0; c < ncomp; ++c)
        if (l <= comps[c].lmax) sqrtS_elem(comps[c], smat, 0, d, nullptr, sx, m, l, false);
}

// ---- the same three updates for ONE diffuse component with ONE map (the T-only headline): identical arithmetic, the
// loops over components and Stokes blocks (runtime trip counts, small arrays indexed at run time) gone.
CMDR_HD bool cg_single(const CompDev* __restrict__ comps, int ncomp) { return ncomp == 1 && comps[0].nmaps == 1; }
// sqrtS_slot for nmaps = 1, kind 0, pass_inactive = false: out = fac * in (+ add where the component has a prior)
CMDR_HD double cg_sqrtS1(const CompDev& C, const double* __restrict__ smat, int l, double v) {
    if (!C.active) return 0.0;
    if (C.lmax_cl < 0) return v;
    if (l > C.lmax_cl) return 0.0;
    double s = 0.0;
    s += smat[C.smat_off + l] * v;
    return s;
}
// ilo <= index < ihi: the slice of the stacked vector this rank owns (m-sliced CG vectors; default: everything).  One
// component with one map: S^1/2 and M^-1 are scalars per l, so the (re, im) slots of an (l, m) are independent and a
// slice may be any contiguous index range.
CMDR_HD double cg_q_elem1(const CompDev& C, const double* __restrict__ smat, const double* __restrict__ yc,
                          const double* __restrict__ d, double* __restrict__ q, int m, int l,
                          int64_t ilo = 0, int64_t ihi = INT64_MAX) {
    const int64_t i0 = C.pos + d_packed_index(C.lmax, l, m);
    const bool addon = C.active && C.lmax_cl >= 0;
    double acc = 0.0;
    for (int sl = 0; sl < (m > 0 ? 2 : 1); ++sl) {
        if (i0 + sl < ilo || i0 + sl >= ihi) continue;
        const double dv = d[i0 + sl];
        const double o = cg_sqrtS1(C, smat, l, yc[i0 + sl]) + (addon ? dv : 0.0);
        q[i0 + sl] = o;
        acc += dv * o;
    }
    return acc;
}
CMDR_HD double cg_xr_elem1(const CompDev& C, const double* __restrict__ P, int lmax_pre, double alpha,
                           double* __restrict__ x, double* __restrict__ r, const double* __restrict__ d,
                           const double* __restrict__ q, double* __restrict__ s, int m, int l,
                           int64_t ilo = 0, int64_t ihi = INT64_MAX) {
    const int64_t i0 = C.pos + d_packed_index(C.lmax, l, m);
    const double p = P[d_moff(lmax_pre, m) + (l - m)];
    double acc = 0.0;
    for (int sl = 0; sl < (m > 0 ? 2 : 1); ++sl) {
        const int64_t i = i0 + sl;
        if (i < ilo || i >= ihi) continue;
        x[i] += alpha * d[i];
        double rn = r[i];
        rn -= alpha * q[i];
        r[i] = rn;
        double sv = 0.0;
        sv += p * rn;
        s[i] = sv;
        acc += rn * sv;
    }
    return acc;
}
CMDR_HD void cg_d_elem1(const CompDev& C, const double* __restrict__ smat, double beta, double* __restrict__ d,
                        const double* __restrict__ s, double* __restrict__ sx, int m, int l,
                        int64_t ilo = 0, int64_t ihi = INT64_MAX) {
    const int64_t i0 = C.pos + d_packed_index(C.lmax, l, m);
    for (int sl = 0; sl < (m > 0 ? 2 : 1); ++sl) {
        if (i0 + sl < ilo || i0 + sl >= ihi) continue;
        const double dn = s[i0 + sl] + beta * d[i0 + sl];
        d[i0 + sl] = dn;
        sx[i0 + sl] = cg_sqrtS1(C, smat, l, dn);
    }
}

}  // namespace cmdr
