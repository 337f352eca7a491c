"""CPU tier, world_size = 2 over gloo: the N > 1 path of bench.py (ring-pair sharding, replicated a_lm, ONE all-reduce
of the partial vector per matvec / RHS through the cmdr_allreduce_fn callback) against the single-rank result.
Kernels run through the host emulation (tests/host_emul); on the GPU box the same host code drives RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

from helpers import ROOT


def _spec(nside, lmax, pix, mode):
    from commander_amd import synth, healpix
    spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, pixels=pix)
    if mode == "varying":   # synchrotron with a spatially varying index + the pseudo-inverse preconditioner
        z = healpix.pix_z(nside)
        if pix is not None:
            z = z[pix]
        spec["comps"][1]["F_map"] = {ib: (b["nu"] / 30.0) ** (-3.1 + 0.1 * z) for ib, b in enumerate(spec["bands"])}
    if mode == "compact":   # templates + point sources (pixel-space components): local rows of the sparse matrices
        synth.add_compact_blocks(spec, nsrc=4)
    return spec


def _md_mask(nside):
    from commander_amd import healpix
    z = healpix.pix_z(nside)
    return ((np.abs(z) > 0.25) * (0.5 + 0.5 * np.random.default_rng(3).random(z.size))).astype(np.float64)


def _worker(rank, world, port, out_dir, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch
    import torch.distributed as dist
    from helpers import emul_lib
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    dist.init_process_group("gloo", rank=rank, world_size=world)
    EL = emul_lib()
    nside, lmax = 16, 32
    rings = healpix.rank_rings(nside, rank, world)
    pix = healpix.local_pixels(nside, rings)
    spec = _spec(nside, lmax, pix, mode)
    ctx = build_context(spec, rings_by_nside={nside: rings}, _lib=EL)

    def allreduce(ptr, n):  # "device" memory is host memory in the emulation
        buf = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(n,))
        t = torch.from_numpy(buf)
        dist.all_reduce(t)
    if mode == "stream":   # the stream-ordered callback variant (RCCL in bench.py); the emulation has no streams
        ctx.set_allreduce_stream(lambda ptr, n, stream: allreduce(ptr, n))
    else:
        ctx.set_allreduce(allreduce)
    ctx.initPrecond("pseudoinv" if mode == "varying" else "diagonal")
    ctx.update_precond()
    x = np.random.default_rng(5).standard_normal(ctx.ncr)
    y = ctx.cr_matmulA(x)
    if mode == "stream":
        # the stream-ordered collective takes the two-half form (adjoint + row-range sums of m < m_split | m >= m_split,
        # the first half's sum on the second stream): same sums in the same order as one launch + one all-reduce
        os.environ["CMDR_OVERLAP"] = "0"
        assert np.array_equal(ctx.cr_matmulA(x), y)
        del os.environ["CMDR_OVERLAP"]
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    d0 = ctx.alpha_nu(0) if mode == "varying" else ctx.invN_diag(0)
    # the 'chisq' criterion sums squared residuals over the ranks' pixels: same stopping iteration as one rank
    solc, nc, statc, _ = ctx.solve_cr_eqn_by_CG(b, "chisq", 1e-3, 2, 60, 1)
    # applyMonoDipolePrior on the ring-sharded map: the masked sums are reduced over the ranks (mpi_allreduce at
    # comm_diffuse_comp_mod.f90:5766-5767, :5792-5793)
    mask = _md_mask(nside)[pix]
    md1, mu1 = ctx.applyMonoDipolePrior(0, sol, nside, mask, "monopole")
    md2, mu2 = ctx.applyMonoDipolePrior(0, sol, nside, mask, "monopole+dipole")
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), y=y, b=b, sol=sol, d0=d0, solc=solc, nc=nc, md1=md1, md2=md2,
             mu1=mu1, mu2=mu2)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["const", "varying", "stream", "compact"])
def test_two_rank_ring_sharding_matches_single_rank(tmp_path, mode):
    import torch.multiprocessing as mp
    from helpers import emul_lib, rel
    from commander_amd import synth
    from commander_amd.cr import build_context
    emul_lib()  # build once before the ranks race for it
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
    EL = emul_lib()
    spec = _spec(16, 32, None, mode)
    ctx = build_context(spec, _lib=EL)
    ctx.initPrecond("pseudoinv" if mode == "varying" else "diagonal")
    ctx.update_precond()
    x = np.random.default_rng(5).standard_normal(ctx.ncr)
    y = ctx.cr_matmulA(x)
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    solc, nc, statc, _ = ctx.solve_cr_eqn_by_CG(b, "chisq", 1e-3, 2, 60, 1)
    md1, mu1 = ctx.applyMonoDipolePrior(0, sol, 16, _md_mask(16), "monopole")
    md2, mu2 = ctx.applyMonoDipolePrior(0, sol, 16, _md_mask(16), "monopole+dipole")
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(g["nc"]) == nc and 2 <= nc < 60 and rel(g["solc"], solc) < 1e-8
        assert rel(g["y"], y) < 1e-12
        assert rel(g["b"], b) < 1e-12
        assert rel(g["d0"], ctx.alpha_nu(0) if mode == "varying" else ctx.invN_diag(0)) < 1e-12
        assert rel(g["sol"], sol) < 1e-10
        # the prior correction of the (slightly different) 2-rank solution equals the single-rank one to the same level
        assert np.allclose(g["mu1"], mu1, rtol=1e-8, atol=1e-10) and np.allclose(g["mu2"], mu2, rtol=1e-8, atol=1e-10)
        assert rel(g["md1"], md1) < 1e-10 and rel(g["md2"], md2) < 1e-10


def _hybrid_worker(rank, world, port, out_dir):
    """4 ranks = 2 band groups x 2 ring sets (band x ring-set hybrid)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch
    import torch.distributed as dist
    from helpers import emul_lib
    from commander_amd import synth, healpix, shard
    from commander_amd.cr import build_context
    dist.init_process_group("gloo", rank=rank, world_size=world)
    EL = emul_lib()
    nside, lmax, nband = 16, 32, 3
    lay = shard.rank_layout(nband, world, rank, band_parts=2, ring_parts=2)
    groups = [dist.new_group([bg * 2, bg * 2 + 1]) for bg in range(2)]      # every rank creates every group
    rings = healpix.rank_rings(nside, lay["ring_index"], lay["ring_parts"])
    pix = healpix.local_pixels(nside, rings)
    spec = synth.make_problem("cfg2", nside=nside, lmax=lmax, pixels=pix, bands=lay["bands"])
    ctx = build_context(spec, rings_by_nside={nside: rings}, _lib=EL)

    def view(ptr, n):
        return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(n,)))
    ctx.set_allreduce(lambda ptr, n: dist.all_reduce(view(ptr, n)))
    ctx.set_band_sharding(lambda ptr, n: dist.all_reduce(view(ptr, n), group=groups[rank // 2]), lay["ring_parts"])
    ctx.initPrecond()
    ctx.update_precond()
    x = np.random.default_rng(5).standard_normal(ctx.ncr)
    y = ctx.cr_matmulA(x)
    pm = ctx.cr_invM(x)
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    # CG_LMAX_PRECOND on top: every rank builds the dense low-l block from ITS bands on the full low-resolution sky, the
    # sum over the band groups counts each group ring_parts times (lowl_update divides by ring_replicas_)
    low = synth.lowres_noise(synth.make_problem("cfg2", nside=nside, lmax=lmax), 4)
    mine = [low[bb] for bb in lay["bands"]]
    ctx.set_lowl_precond(0, 6, [ns for ns, _ in mine], [mm for _, mm in mine])
    ctx.update_precond()
    pml = ctx.cr_invM(x)
    soll = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)[0]
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), y=y, b=b, sol=sol, pm=pm, pml=pml, soll=soll, bands=np.array(lay["bands"]))
    dist.barrier()
    dist.destroy_process_group()


def test_four_rank_band_ring_hybrid_matches_single_rank(tmp_path):
    import torch.multiprocessing as mp
    from helpers import emul_lib, rel
    from commander_amd import synth, shard
    from commander_amd.cr import build_context
    assert shard.plan_shards(9, 8) == (1, 8) and shard.plan_shards(9, 16) == (2, 8) and shard.plan_shards(9, 4) == (1, 4) and shard.plan_shards(9, 1) == (1, 1)
    emul_lib()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_hybrid_worker, args=(4, port, str(tmp_path)), nprocs=4, join=True)
    EL = emul_lib()
    spec = synth.make_problem("cfg2", nside=16, lmax=32)
    ctx = build_context(spec, _lib=EL)
    ctx.initPrecond()
    ctx.update_precond()
    x = np.random.default_rng(5).standard_normal(ctx.ncr)
    y, pm = ctx.cr_matmulA(x), ctx.cr_invM(x)
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    low = synth.lowres_noise(spec, 4)
    ctx.set_lowl_precond(0, 6, [ns for ns, _ in low], [mm for _, mm in low])
    ctx.update_precond()
    pml = ctx.cr_invM(x)
    soll = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)[0]
    assert np.count_nonzero(pml != pm) == 49                   # the (6 + 1)^2 low-l entries of the CMB went through the dense block
    seen = []
    for r in range(4):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        seen += list(g["bands"])
        assert rel(g["y"], y) < 1e-12
        assert rel(g["pm"], pm) < 1e-12
        assert rel(g["b"], b) < 1e-12
        assert rel(g["sol"], sol) < 1e-10
        assert rel(g["pml"], pml) < 1e-11                      # band x ring-set bookkeeping of the low-l block (ring_replicas = 2)
        assert rel(g["soll"], soll) < 1e-10
    assert sorted(set(seen)) == [0, 1, 2]


def _slice_worker(rank, world, port, out_dir, kind):
    """m-sliced CG vectors (cmdr_ctx_set_vector_slicing) on top of ring sharding: one scalar component, nine bands."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch
    import torch.distributed as dist
    from helpers import emul_lib
    from commander_amd import synth, healpix
    from commander_amd.cr import build_context
    dist.init_process_group("gloo", rank=rank, world_size=world)
    EL = emul_lib()
    nside, lmax = 16, 32
    rings = healpix.rank_rings(nside, rank, world)
    pix = healpix.local_pixels(nside, rings)
    spec = synth.make_problem("cfg3", nside=nside, lmax=lmax, pixels=pix)
    ctx = build_context(spec, rings_by_nside={nside: rings}, _lib=EL)
    calls = []

    def allreduce(ptr, n):
        calls.append(n)
        dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(n,))))
    if kind == "stream":
        ctx.set_allreduce_stream(lambda ptr, n, stream: allreduce(ptr, n))
    else:
        ctx.set_allreduce(allreduce)
    ctx.initPrecond()
    ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    plain, n0, _, res0 = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    ctx.set_vector_slicing(rank, world)
    del calls[:]
    sl, n1, _, res1 = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    small = [c for c in calls if c < 64]                 # the dot products' block partials (1 entry in the emulation)
    slr, nr, statr, _ = ctx.solve_cr_eqn_by_CG(b, "residual", 1e-10, 2, 200, 1)
    ctx.set_vector_slicing(0, 1)
    plr, npr, _, _ = ctx.solve_cr_eqn_by_CG(b, "residual", 1e-10, 2, 200, 1)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), plain=plain, sl=sl, res0=res0, res1=res1, nsmall=len(small),
             slr=slr, plr=plr, nr=nr, npr=npr, x0=ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 6, 1, x0=plain)[0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "blocking"), (2, "stream"), (4, "blocking")])
def test_sliced_cg_vectors_match_replicated_vectors(tmp_path, world, kind):
    """with cmdr_ctx_set_vector_slicing every rank keeps 1 / P of x, r, d, q, s inside the PCG loop
    (reduce-scatter of the matvec output, all-gather of S^1/2 d, block partials of the two dot products summed over the
    ranks); the solution must equal the replicated-vector solve to rounding, on every rank, for fixed_iter and for the
    residual criterion (same iteration count)."""
    import torch.multiprocessing as mp
    from helpers import emul_lib, rel
    emul_lib()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_slice_worker, args=(world, port, str(tmp_path), kind), nprocs=world, join=True)
    g0 = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert rel(g["sl"], g["plain"]) < 1e-11 and not np.array_equal(g["sl"], g["plain"])   # other summation grouping
        assert np.allclose(g["res1"], g["res0"], rtol=1e-9)
        assert int(g["nsmall"]) == 2 * 12                       # two dot-product sums per iteration
        assert int(g["nr"]) == int(g["npr"]) and rel(g["slr"], g["plr"]) < 1e-9
        assert np.array_equal(g["sl"], g0["sl"])                # every rank ends with the same gathered solution
