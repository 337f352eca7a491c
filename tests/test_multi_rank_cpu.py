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
    return spec


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
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol, n, stat, res = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", 1e-8, 5, 12, 1)
    d0 = ctx.alpha_nu(0) if mode == "varying" else ctx.invN_diag(0)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), y=y, b=b, sol=sol, d0=d0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["const", "varying", "stream"])
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
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert rel(g["y"], y) < 1e-12
        assert rel(g["b"], b) < 1e-12
        assert rel(g["d0"], ctx.alpha_nu(0) if mode == "varying" else ctx.invN_diag(0)) < 1e-12
        assert rel(g["sol"], sol) < 1e-10
