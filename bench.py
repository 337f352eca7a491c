#!/usr/bin/env python3
"""bench.py -- Gibbs amplitude-sample CG solves/sec on MI355X (BASELINE.json metric).

One "step" = one `sample_amps_by_CG`-equivalent (commander3/src/comm_signal_mod.f90:154-216):
cr_computeRHS ('sample') + solve_cr_eqn_by_CG with the shipped settings (fixed_iter, 40 iterations, diagonal
preconditioner; commander3/parameter_files/param_BP_v8.00_full.txt:40-47,782) on BASELINE.json configs[2]'s
geometry: 9 Planck-like bands, CMB, Nside=1024, lmax=2000, fp64, synthetic inputs resident in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): HEALPix ring pairs are dealt round-robin to the ranks
(Commander's own pixel distribution, comm_map_mod.f90:197-221) for every band, harmonic-space vectors stay
replicated, and the per-matvec partial vector is summed with one RCCL all-reduce (torch.distributed backend
"nccl").  Total work is fixed => "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 matrix peak (guide has no fp64 row)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NITER = 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class CudaView:
    """Zero-copy view of a raw device pointer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def cpu_baseline(spec, nthreads):
    """Reported, not targeted: the CPU oracle (C + OpenMP restatement, oracle/sht_oracle.c) timed on this box's
    host cores on a bounded sample -- one Yt+Y pair at the benchmark geometry -- and extrapolated to a solve
    (9 bands x (41 matvecs + RHS) pairs).  It is NOT the Fortran+libsharp2 binary (unbuildable here)."""
    from oracle import sht
    sht.build()
    nside, lmax = spec["nside"], spec["lmax"]
    rng = np.random.default_rng(0)
    m = rng.standard_normal(12 * nside * nside)
    t0 = time.time()
    a = sht.Yt(nside, lmax, m, nthreads=nthreads)
    y = sht.Y(nside, lmax, a, nthreads=nthreads)
    t_pair = time.time() - t0
    reps = 1
    while t_pair * reps < 10.0 and reps < 4:   # keep the sample to ~10-30 s of CPU work
        t0 = time.time()
        a = sht.Yt(nside, lmax, y, nthreads=nthreads)
        y = sht.Y(nside, lmax, a, nthreads=nthreads)
        t_pair = min(t_pair, time.time() - t0)
        reps += 1
    nb = len(spec["bands"])
    pairs_per_solve = nb * (NITER + 1) + nb * 0.5
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / (pairs_per_solve * t_pair), "unit": "solves/s", "cores": nthreads, "kind": "port",
            "cpu": cpu,
            "sample": "best of %d Yt+Y pairs (oracle/sht_oracle.c, OpenMP) at Nside=%d lmax=%d = %.2f s each; "
                      "extrapolated x %.1f pairs per solve" % (reps, nside, lmax, t_pair, pairs_per_solve)}


def sht_pairs(L, nside, lmax, reps=10, pols=(False, True)):
    import ctypes as C
    from commander_amd.lib import check
    res = {}
    for pol in pols:
        h = C.c_void_p()
        create = L.cmdr_sht_plan_create_pol if pol else L.cmdr_sht_plan_create
        check(create(nside, lmax, 0, None, None, 2 if pol else 1, C.byref(h)), L)
        na, npx = L.cmdr_sht_nalm(h), L.cmdr_sht_npix(h)
        bufs = []
        for n in (na, na, npx, npx):
            p = C.c_void_p()
            check(L.cmdr_dev_alloc(n * 8, C.byref(p)), L)
            a = np.random.default_rng(n).standard_normal(n)
            check(L.cmdr_memcpy_h2d(p, a.ctypes.data_as(C.c_void_p), a.nbytes), L)
            bufs.append(p)
        dE, dB, dQ, dU = bufs

        def pair():
            if pol:
                check(L.cmdr_sht_execute_spin2_dev(h, 2, dE, dB, dQ, dU), L)   # Yt
                check(L.cmdr_sht_execute_spin2_dev(h, 1, dE, dB, dQ, dU), L)   # Y
            else:
                check(L.cmdr_sht_execute_dev(h, 2, 1, dE, na, dQ, npx), L)
                check(L.cmdr_sht_execute_dev(h, 1, 1, dE, na, dQ, npx), L)
        pair()
        t0 = time.perf_counter()
        for _ in range(reps):
            pair()
        dt = (time.perf_counter() - t0) / reps
        res["spin2_QU" if pol else "spin0"] = 1.0 / dt
        for p in bufs:
            L.cmdr_dev_free(p)
        L.cmdr_sht_plan_destroy(h)
    res["geometry"] = "Nside=%d lmax=%d" % (nside, lmax)
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dist = os.environ.get("CMDR_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path on a 1-GPU box
    # rehearsal of the N > 1 logic on a 1-GPU box: every rank on device 0, collectives through gloo on host copies
    one_gpu = os.environ.get("CMDR_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="gloo" if one_gpu else "nccl", rank=rank, world_size=world)
    from commander_amd import synth, healpix, shard
    from commander_amd.cr import build_context

    cfg = args.config
    nside, lmax = synth.CONFIGS[cfg]["nside"], synth.CONFIGS[cfg]["lmax"]
    rings = pixels = bands = None
    lay = shard.rank_layout(len(synth.CONFIGS[cfg]["nu"]), world, rank)
    if os.environ.get("CMDR_BENCH_SHARD"):          # "BxR": force band_parts x ring_parts
        bp, rp = (int(v) for v in os.environ["CMDR_BENCH_SHARD"].split("x"))
        lay = shard.rank_layout(len(synth.CONFIGS[cfg]["nu"]), world, rank, bp, rp)
    ring_groups = None
    if world > 1:
        # band x ring-set hybrid: this rank owns lay["bands"] on ring set lay["ring_index"]
        if lay["ring_parts"] > 1:
            rings = healpix.rank_rings(nside, lay["ring_index"], lay["ring_parts"])
            pixels = healpix.local_pixels(nside, rings)
        if lay["band_parts"] > 1:
            bands = lay["bands"]
            if lay["ring_parts"] > 1:   # every rank creates every group, in the same order
                ring_groups = [dist.new_group([bg * lay["ring_parts"] + i for i in range(lay["ring_parts"])])
                               for bg in range(lay["band_parts"])]
    spec = synth.make_problem(cfg, pixels=pixels, bands=bands)
    ctx = build_context(spec, device=local_rank, rings_by_nside={nside: rings} if rings is not None else None)
    if dist is not None:
        import torch

        views = {}

        def allreduce(ptr, n, stream):
            # RCCL all-reduce ordered on the library's own stream: torch makes its NCCL stream wait for the current
            # stream before the collective and the current stream wait for the collective after it, so nothing here
            # blocks the host and a whole fixed_iter solve stays queued ahead of the GPU
            key = (ptr, n)
            if key not in views:
                views[key] = torch.as_tensor(CudaView(ptr, n), device="cuda:%d" % local_rank)
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device="cuda:%d" % local_rank)):
                dist.all_reduce(views[key])
        if world > 1 and lay["band_parts"] > 1:
            grp = ring_groups[rank // lay["ring_parts"]] if ring_groups else None

            def allreduce_rings(ptr, n):   # setup-time only (noise a_lm of a band over its ring sets)
                t = torch.as_tensor(CudaView(ptr, n), device="cuda:%d" % local_rank)
                if one_gpu:
                    h = t.cpu()
                    dist.all_reduce(h, group=grp)
                    t.copy_(h)
                else:
                    dist.all_reduce(t, group=grp)
                torch.cuda.synchronize()
            ctx.set_band_sharding(allreduce_rings if grp is not None else None, lay["ring_parts"])
        if os.environ.get("CMDR_BENCH_BLOCKING_ALLREDUCE") == "1" or one_gpu:   # the MPI-style blocking callback
            def allreduce_blocking(ptr, n):
                t = torch.as_tensor(CudaView(ptr, n), device="cuda:%d" % local_rank)
                if one_gpu:
                    h = t.cpu()
                    dist.all_reduce(h)
                    t.copy_(h)
                else:
                    dist.all_reduce(t)
                torch.cuda.synchronize()
            ctx.set_allreduce(allreduce_blocking)
        else:
            ctx.set_allreduce_stream(allreduce)
    ctx.initPrecond()
    ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    dres = [ctx.dev(r.size, r) for r in resid]
    dxi = [ctx.dev(r.size, r) for r in xi]
    deta = ctx.dev(ctx.ncr, eta)
    b, x = ctx.dev(ctx.ncr), ctx.dev(ctx.ncr)
    L = ctx.L

    def step():
        ctx.cr_computeRHS_dev("sample", dres, dxi, deta, None, b)
        return ctx.solve_dev(b, x, "fixed_iter", 1e-8, 5, NITER, 1)

    def barrier():
        L.cmdr_device_synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    L.cmdr_profile_enable(ctx._h, 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        niter, stat, res = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda:%d" % local_rank)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = (ctypes.c_double * 4)()
    cnt = (ctypes.c_longlong * 4)()
    L.cmdr_profile_read(ctx._h, ms, cnt)
    info = (ctypes.c_int64 * 3)()
    L.cmdr_problem_info(ctx._h, info)
    if rank == 0:
        nbm, steps_per_map = int(info[0]), int(info[2])
        # dominant kernel: the Legendre stage (synthesis + adjoint launches, same algorithmic work each):
        # 8 flop per (ring pair, l, m) recursion step (2 FMA recursion + 2 FMA accumulate), on the
        # mlim-pruned steps this rank actually owns, x the (band, Stokes) maps one launch processes.
        flop_launch = 8.0 * steps_per_map * nbm
        nl = int(cnt[0] + cnt[2])
        t_leg = (ms[0] + ms[2]) / max(nl, 1) * 1e-3
        achieved = flop_launch / t_leg / 1e12 if nl else 0.0
        # whole-iteration HBM view (SURVEY.md §8d B_iter): per band 8(2 nalm + 3 npix) + 10*8*ncr
        npix_loc = sum(s[0] for s in ctx.band_shape)
        b_iter = 8.0 * (2 * (lmax + 1) ** 2 * len(spec["bands"]) + 3 * npix_loc) + 80.0 * ctx.ncr
        t_mv = ms[3] / max(int(cnt[3]), 1) * 1e-3
        traffic = None   # HBM bytes per Legendre launch from separate rocprofv3 --pmc passes (profiles/), if present
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            if world == 1 and cfg == "cfg3":
                traffic = tj["legendre_span_bytes"]["mean"]
        except Exception:
            pass
        out = {
            "metric": "cg_solves_per_sec", "value": args.steps / dt, "unit": "solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: 9 Planck-like bands, CMB T-only, Nside=%d lmax=%d, "
                                   "amp-sample = cr_computeRHS + %d fixed PCG iterations, diagonal preconditioner"
                                   % (nside, lmax, NITER),
                       "parallelism": ("%d band groups x %d ring sets (hybrid sharding)" % (lay["band_parts"], lay["ring_parts"])
                                       if lay["band_parts"] > 1 else "ring-pair sharding x%d" % world)
                                      + ", replicated a_lm, 1 all-reduce(ncr) per matvec"
                       if world > 1 else "single GPU", "ncr": ctx.ncr, "cg_iters_per_sec": args.steps * NITER / dt},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_PEAK_TFLOPS, "traffic": traffic,
                         "kernel": "k_leg_synth + k_leg_adj (fp64 Legendre stage; VALU-bound, MI355X fp64 matrix "
                                   "peak equals the vector peak)",
                         "avg_launch_ms": t_leg * 1e3, "launches": nl, "flop_per_launch": flop_launch,
                         "hbm_iter_gbs": b_iter / t_mv / 1e9 if t_mv else None,
                         # whole-matvec fp64 view (SURVEY.md 8d F_iter, mlim-pruned): synthesis + adjoint flops over
                         # the matvec wall time, ring stage and streams included
                         "fp64_iter_tflops": 2.0 * flop_launch / t_mv / 1e12 if t_mv else None,
                         "hbm_iter_frac": b_iter / t_mv / 1e9 / HBM_PEAK_GBS if t_mv else None,
                         "ms": {"leg_synth": ms[0] / max(int(cnt[0]), 1), "ring_fused": ms[1] / max(int(cnt[1]), 1),
                                "leg_adjoint": ms[2] / max(int(cnt[2]), 1), "matvec": ms[3] / max(int(cnt[3]), 1)}},
            "solve": {"niter": niter, "stat": stat, "res": res[0], "delta0": res[1]},
        }
        # second half of the headline metric: SHT pairs/s/GPU, the unit of commander3/src/sharp_test.f90:65-71
        # (one Yt followed by one Y), scalar and polarised, at the benchmark geometry, data resident in HBM
        try:
            out["sht_pairs_per_sec_per_gpu"] = sht_pairs(L, nside, lmax) if world == 1 else None
            if world == 1:   # the reference's own SHT benchmark geometry, commander3/src/sharp_test.f90:31-33
                out["sht_pairs_per_sec_per_gpu"]["sharp_test"] = sht_pairs(L, 2048, 3 * 2048, reps=5, pols=(False,))
        except Exception as e:
            out["sht_pairs_per_sec_per_gpu"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(spec, min(os.cpu_count() or 1, 16))
            except Exception as e:  # the baseline leg must never sink the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
