"""The literal libsharp2 symbols of include/cmdr_sharp.h (what commander3/src/sharp.f90 binds), driven the way
``sharp_execute_d`` drives them (sharp.f90:186-241): info objects, array-of-column-pointers, job codes."""
import ctypes

import numpy as np
import pytest

from helpers import emul_lib, rel

SHARP_DP = 1 << 4
dp = ctypes.POINTER(ctypes.c_double)


def run_sharp(L, oracle_lib, nside, lmax, P=1, rank=0):
    vp = ctypes.c_void_p
    L.sharp_alm_count.restype = ctypes.c_ssize_t
    L.sharp_alm_count.argtypes = [vp]
    L.sharp_map_size.restype = ctypes.c_ssize_t
    L.sharp_map_size.argtypes = [vp]
    L.sharp_make_mmajor_real_packed_alm_info.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]
    L.sharp_make_subset_healpix_geom_info.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), dp, ctypes.POINTER(vp)]
    L.sharp_execute_mpi_fortran.argtypes = [ctypes.c_int] * 3 + [vp, vp, vp, vp, ctypes.c_int, dp, ctypes.POINTER(ctypes.c_ulonglong)]
    L.sharp_destroy_alm_info.argtypes = [vp]
    L.sharp_destroy_geom_info.argtypes = [vp]
    from oracle import healpix
    rng = np.random.default_rng(nside + lmax)
    ms = np.arange(lmax + 1, dtype=np.int32)
    ainfo, ginfo = vp(), vp()
    L.sharp_make_mmajor_real_packed_alm_info(lmax, 1, lmax + 1, ms.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(ainfo))
    assert L.sharp_alm_count(ainfo) == (lmax + 1) ** 2
    north = np.arange(1 + rank, 2 * nside + 1, P)
    rings = np.array(sorted(list(north) + [4 * nside - i for i in north if i < 2 * nside]), dtype=np.int32)
    w = 1.0 + 0.05 * rng.standard_normal(2 * nside)
    L.sharp_make_subset_healpix_geom_info(nside, 1, rings.size, rings.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                          w.ctypes.data_as(dp), ctypes.byref(ginfo))
    idx = np.concatenate([np.arange(healpix.ring_info(nside, i)[4], healpix.ring_info(nside, i)[4] + healpix.ring_info(nside, i)[0]) for i in rings])
    assert L.sharp_map_size(ginfo) == idx.size
    a = rng.standard_normal((lmax + 1) ** 2)
    m = np.zeros(idx.size)
    pa = (dp * 1)(a.ctypes.data_as(dp))
    pm = (dp * 1)(m.ctypes.data_as(dp))
    t = ctypes.c_double(0)
    L.sharp_execute_mpi_fortran(0, 1, 0, pa, pm, ginfo, ainfo, SHARP_DP, ctypes.byref(t), None)     # SHARP_Y
    assert rel(m, oracle_lib.Y(nside, lmax, a)[idx]) < 1e-11
    if P == 1:
        m2 = rng.standard_normal(idx.size)
        out = np.zeros_like(a)
        pa = (dp * 1)(out.ctypes.data_as(dp))
        pm = (dp * 1)(m2.ctypes.data_as(dp))
        L.sharp_execute_mpi_fortran(0, 0, 0, pa, pm, ginfo, ainfo, SHARP_DP, None, None)            # SHARP_YtW
        assert rel(out, oracle_lib.YtW(nside, lmax, m2, wring=w)) < 1e-11
        # spin 2: alm = (E, B), map = (Q, U)  (comm_map_mod.f90:446-449)
        e, b = rng.standard_normal((lmax + 1) ** 2), rng.standard_normal((lmax + 1) ** 2)
        q, u = np.zeros(idx.size), np.zeros(idx.size)
        pa = (dp * 2)(e.ctypes.data_as(dp), b.ctypes.data_as(dp))
        pm = (dp * 2)(q.ctypes.data_as(dp), u.ctypes.data_as(dp))
        L.sharp_execute_mpi_fortran(0, 1, 2, pa, pm, ginfo, ainfo, SHARP_DP, None, None)
        qo, uo = oracle_lib.sht_spin2(1, nside, lmax, almE=e, almB=b)
        assert rel(np.concatenate([q, u]), np.concatenate([qo, uo])) < 1e-11
    L.sharp_destroy_alm_info(ainfo)
    L.sharp_destroy_geom_info(ginfo)


def test_sharp_symbols_host_logic(oracle_lib):
    EL = emul_lib()
    run_sharp(EL, oracle_lib, 8, 16)
    run_sharp(EL, oracle_lib, 8, 16, P=2, rank=1)


@pytest.mark.gpu
def test_sharp_symbols_gpu(oracle_lib):
    from commander_amd import get_lib
    run_sharp(get_lib(), oracle_lib, 64, 128)
    run_sharp(get_lib(), oracle_lib, 32, 64, P=3, rank=2)


# ---- P = 2 ranks, both rings AND m distributed, exactly Commander's ownership (comm_map_mod.f90:193-261) ------------
ALLRED = ctypes.CFUNCTYPE(None, ctypes.c_void_p, dp, ctypes.c_longlong)


def _local_alm_index(lmax, ms):
    """positions in the full (all-m) packed vector of a rank's local a_lm, in its own order (ms order, l inner)."""
    idx = []
    for m in ms:
        start = 0 if m == 0 else 2 * (m * (lmax + 1) - m * (m - 1) // 2) - (lmax + 1)
        idx += list(range(start, start + (lmax + 1 if m == 0 else 2 * (lmax + 1 - m))))
    return np.array(idx)


def _sharp_rank(rank, world, port, out_dir, nside, lmax, gpu=False):
    import os
    import sys
    from helpers import ROOT
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import healpix
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if gpu:     # the PRODUCT library: both ranks on device 0, the a_lm sum over the "communicator" through gloo on host buffers
        from commander_amd import get_lib
        L = get_lib()
    else:
        L = emul_lib()
    vp = ctypes.c_void_p
    L.sharp_alm_count.restype = ctypes.c_ssize_t
    L.sharp_alm_count.argtypes = [vp]
    L.sharp_map_size.restype = ctypes.c_ssize_t
    L.sharp_map_size.argtypes = [vp]
    L.sharp_make_mmajor_real_packed_alm_info.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp)]
    L.sharp_make_subset_healpix_geom_info.argtypes = [ctypes.c_int] * 3 + [ctypes.POINTER(ctypes.c_int), dp, ctypes.POINTER(vp)]
    L.sharp_execute_mpi_fortran.argtypes = [ctypes.c_int] * 3 + [vp, vp, vp, vp, ctypes.c_int, dp, ctypes.POINTER(ctypes.c_ulonglong)]
    L.cmdr_sharp_register_comm.argtypes = [ctypes.c_int, ALLRED, vp]

    def allreduce(user, buf, n):          # what a 6-line MPI_Allreduce(MPI_IN_PLACE, ...) wrapper does in the Fortran driver
        dist.all_reduce(torch.from_numpy(np.ctypeslib.as_array(buf, shape=(n,))))
    cb = ALLRED(allreduce)
    COMM = 7                               # any handle value: the Fortran MPI communicator (MPI_Fint)
    L.cmdr_sharp_register_comm(COMM, cb, None)
    ms = np.arange(rank, lmax + 1, world, dtype=np.int32)              # comm_map_mod.f90:231
    north = np.arange(1 + rank, 2 * nside + 1, world)                  # :197
    rings = np.array(sorted(list(north) + [4 * nside - i for i in north if i < 2 * nside]), dtype=np.int32)
    ainfo, ginfo = vp(), vp()
    L.sharp_make_mmajor_real_packed_alm_info(lmax, 1, ms.size, ms.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(ainfo))
    w = 1.0 + 0.05 * np.random.default_rng(3).standard_normal(2 * nside)
    L.sharp_make_subset_healpix_geom_info(nside, 1, rings.size, rings.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                          w.ctypes.data_as(dp), ctypes.byref(ginfo))
    ai = _local_alm_index(lmax, ms)
    pix = np.concatenate([np.arange(healpix.ring_info(nside, i)[4], healpix.ring_info(nside, i)[4] + healpix.ring_info(nside, i)[0]) for i in rings])
    assert L.sharp_alm_count(ainfo) == ai.size and L.sharp_map_size(ginfo) == pix.size
    g = np.random.default_rng(11)                                      # the same global fields on every rank
    na, npx = (lmax + 1) ** 2, 12 * nside * nside
    a_full, e_full, b_full = g.standard_normal(na), g.standard_normal(na), g.standard_normal(na)
    m_full, q_full, u_full = g.standard_normal(npx), g.standard_normal(npx), g.standard_normal(npx)
    out = {}

    def col(v):
        return np.ascontiguousarray(v)

    def run(job, spin, alms, maps):
        pa = (dp * len(alms))(*[x.ctypes.data_as(dp) for x in alms])
        pm = (dp * len(maps))(*[x.ctypes.data_as(dp) for x in maps])
        L.sharp_execute_mpi_fortran(COMM, job, spin, pa, pm, ginfo, ainfo, SHARP_DP, None, None)
    a_loc, m_loc = col(a_full[ai]), np.zeros(pix.size)
    run(1, 0, [a_loc], [m_loc]); out["Y"] = m_loc.copy()               # SHARP_Y
    run(3, 0, [a_loc], [m_loc]); out["WY"] = m_loc.copy()              # SHARP_WY
    m_in, a_out = col(m_full[pix]), np.zeros(ai.size)
    run(2, 0, [a_out], [m_in]); out["Yt"] = a_out.copy()               # SHARP_Yt
    run(0, 0, [a_out], [m_in]); out["YtW"] = a_out.copy()              # SHARP_YtW
    e_loc, b_loc, q_loc, u_loc = col(e_full[ai]), col(b_full[ai]), np.zeros(pix.size), np.zeros(pix.size)
    run(1, 2, [e_loc, b_loc], [q_loc, u_loc]); out["Y2"] = np.concatenate([q_loc, u_loc])
    q_in, u_in, e_out, b_out = col(q_full[pix]), col(u_full[pix]), np.zeros(ai.size), np.zeros(ai.size)
    run(2, 2, [e_out, b_out], [q_in, u_in]); out["Yt2"] = np.concatenate([e_out, b_out])
    np.savez(os.path.join(out_dir, "sharp%d.npz" % rank), ai=ai, pix=pix, **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharp_two_ranks_rings_and_m_distributed_gpu(tmp_path, oracle_lib):
    """The same with the product library on the GPU: two fresh processes on device 0."""
    _two_rank_sharp(tmp_path, oracle_lib, 32, 64, True)


def test_sharp_two_ranks_rings_and_m_distributed(tmp_path, oracle_lib):
    """VERDICT r1 item 6 / SURVEY 8(a14): the nine symbols driven as sharp.f90:186-241 does with P = 2 -- each rank owns
    every second ring pair and every second m -- give each rank its slice of the one-rank result."""
    _two_rank_sharp(tmp_path, oracle_lib, 8, 16, False)


def _two_rank_sharp(tmp_path, oracle_lib, nside, lmax, gpu):
    import os
    import socket
    import torch.multiprocessing as mp
    from oracle import sht as osht
    if not gpu:
        emul_lib()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sharp_rank, args=(2, port, str(tmp_path), nside, lmax, gpu), nprocs=2, join=True)
    g = np.random.default_rng(11)
    na, npx = (lmax + 1) ** 2, 12 * nside * nside
    a_full, e_full, b_full = g.standard_normal(na), g.standard_normal(na), g.standard_normal(na)
    m_full, q_full, u_full = g.standard_normal(npx), g.standard_normal(npx), g.standard_normal(npx)
    w = 1.0 + 0.05 * np.random.default_rng(3).standard_normal(2 * nside)
    ref = dict(Y=oracle_lib.Y(nside, lmax, a_full), WY=oracle_lib.WY(nside, lmax, a_full, wring=w),
               Yt=oracle_lib.Yt(nside, lmax, m_full), YtW=oracle_lib.YtW(nside, lmax, m_full, wring=w))
    q, u = osht.sht_spin2(1, nside, lmax, almE=e_full, almB=b_full)
    e, b = osht.sht_spin2(2, nside, lmax, mapQ=q_full, mapU=u_full)
    seen_a, seen_p = [], []
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), "sharp%d.npz" % r))
        ai, pix = d["ai"], d["pix"]
        seen_a += list(ai); seen_p += list(pix)
        assert rel(d["Y"], ref["Y"][pix]) < 1e-11 and rel(d["WY"], ref["WY"][pix]) < 1e-11
        assert rel(d["Yt"], ref["Yt"][ai]) < 1e-11 and rel(d["YtW"], ref["YtW"][ai]) < 1e-11
        assert rel(d["Y2"], np.concatenate([q[pix], u[pix]])) < 1e-11
        assert rel(d["Yt2"], np.concatenate([e[ai], b[ai]])) < 1e-11
    assert sorted(seen_a) == list(range(na)) and sorted(seen_p) == list(range(npx))   # the two ranks tile everything
