#!/usr/bin/env python3
"""bench.py -- Gibbs amplitude-sample CG solves/sec on MI355X (BASELINE.json metric).

One "step" = one `sample_amps_by_CG`-equivalent (commander3/src/comm_signal_mod.f90:154-216):
cr_computeRHS ('sample') + solve_cr_eqn_by_CG with the shipped settings (fixed_iter, 40 iterations, diagonal
preconditioner; commander3/parameter_files/param_BP_v8.00_full.txt:40-47,782) on BASELINE.json configs[2]'s
geometry: 9 Planck-like bands, CMB, Nside=1024, lmax=2000, fp64, synthetic inputs resident in HBM.

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF:
this (parent) process never touches the GPU -- it only spawns N fresh children with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT set, forwards rank 0's JSON line and exits with the worst child status.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the environment is already there and the
process is a rank.  A WORLD_SIZE that disagrees with --gpus is an error, never a silent 1-GPU run.

N > 1: one rank per GPU.  HEALPix ring pairs (Commander's own pixel distribution, comm_map_mod.f90:197-221) and,
at 8 ranks, band groups are dealt to the ranks (commander_amd/shard.py: band x ring-set hybrid);
harmonic-space vectors stay replicated and the per-matvec partial vector is summed with ONE RCCL all-reduce over
xGMI, issued by the library itself on its own HIP stream (cmdr_ctx_init_rccl; the ncclUniqueId travels through
torch.distributed, which also provides the barrier and the max-over-ranks of the timings).  Total work is fixed
=> "scaling": "strong".

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == fp64 matrix peak (MI355X_MICROARCH.md has no fp64 row; AMD CDNA4 spec)
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
NITER = 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--nside", type=int, default=None, help="rehearsal sizes only; the bench line is the default")
    ap.add_argument("--lmax", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the SHT-pairs and precond-refresh legs")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n):
    """Parent of an N-rank run: spawn N fresh rank processes (this process has not initialised HIP / torch.cuda and
    never will), relay rank 0's stdout, exit with the worst status.  No exec of a GPU process, no re-exec."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Supervise: rank 0's stdout is drained by a thread while every child is polled.  The first rank that exits non-zero
    # (or the wall-clock limit) ends the run: the others would sit in a collective until a watchdog fires, holding
    # their GPUs.  They are fresh children of this process -- terminated here, never re-launched (that is the caller's call).
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("CMDR_BENCH_LAUNCH_TIMEOUT", "3000"))
    t0 = time.time()
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(rcs) if c not in (None, 0)]
        if bad:
            failed = "ranks failed: %r" % (bad,)
            break
        if all(c == 0 for c in rcs):
            break
        if time.time() - t0 > limit:
            failed = "no result after %.0f s (CMDR_BENCH_LAUNCH_TIMEOUT)" % limit
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t1 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: %s; remaining ranks terminated\n" % failed)
        sys.exit(1)
    sys.exit(0)


class CudaView:
    """Zero-copy view of a raw device pointer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def host_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota when one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(cfg, nside, lmax):
    """Reported, not targeted: the CPU oracle (numpy + C/OpenMP restatement under oracle/) timed on this box's host
    cores on a bounded sample of the SAME workload: one real `cr_matmulA` of the oracle on the first bands of the
    benchmark problem (all nine when the cores allow), scaled to nine bands and to a solve (RHS ~ half a matvec +
    41 matvecs: r = b - A x0 is skipped for x0 = 0, 40 iterations + the M^-1 applications are negligible).  It is
    NOT the Fortran+libsharp2 binary (unbuildable here: no HEALPix/libsharp2/FFTW/gfortran)."""
    import numpy as np
    from commander_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_system
    cores = host_cores()
    os.environ["ORACLE_THREADS"] = str(cores)
    nb_all = len(synth.CONFIGS[cfg]["nu"])
    nb = nb_all if cores >= 48 else min(nb_all, 3)
    spec = synth.make_problem(cfg, nside=nside, lmax=lmax, bands=list(range(nb)))
    S = oracle_system(spec)
    x = np.random.default_rng(0).standard_normal(S.ncr)
    t0 = time.time()
    S.matmulA(x)
    t_s = time.time() - t0
    t_mv = t_s * nb_all / nb
    mv_per_solve = NITER + 1 + 0.5
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": 1.0 / (mv_per_solve * t_mv), "unit": "solves/s", "cores": cores, "kind": "port", "cpu": cpu,
            "matvec_s": t_mv,
            "sample": "one oracle cr_matmulA (oracle/cr_oracle.py + oracle/sht_oracle.c, OpenMP x%d) on %d of the %d "
                      "bands at Nside=%d lmax=%d: %.1f s; x %d/%d bands x %.1f matvec-equivalents per solve"
                      % (cores, nb, nb_all, nside, lmax, t_s, nb_all, nb, mv_per_solve)}


def sht_pairs(L, nside, lmax, reps=10, pols=(False, True)):
    import ctypes as C
    import numpy as np
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


def profile_summary(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        launch_ranks(args.gpus)          # does not return
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to measure a different rank count than "
                         "asked for\n" % (args.gpus, world))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import numpy as np
    dist = None
    force_dist = os.environ.get("CMDR_BENCH_FORCE_DIST") == "1"   # exercise the RCCL path on a 1-GPU box
    # rehearsal of the N > 1 logic on a 1-GPU box: every rank on device 0, collectives through gloo on host copies
    one_gpu = os.environ.get("CMDR_BENCH_ONE_GPU") == "1"
    collective = os.environ.get("CMDR_BENCH_COLLECTIVE", "rccl-native")   # | torch-stream | blocking
    # CPU rehearsal of the launch / sharding / JSON logic through the host emulation of the library (tests/host_emul,
    # TEST INFRASTRUCTURE; tests/test_multi_rank_cpu.py): never a measurement, and labelled as such in the output
    rehearse = os.environ.get("CMDR_BENCH_REHEARSE_EMUL") == "1"
    if rehearse:
        one_gpu = True
    if one_gpu:
        local_rank = 0
        collective = "blocking"
    if world > 1 or force_dist:
        import torch
        import torch.distributed as dist
        if not rehearse:
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo" if one_gpu else "nccl", rank=rank, world_size=world)
        if os.environ.get("CMDR_BENCH_TEST_DIE_RANK") == str(rank):   # test hook: a rank that dies during setup
            os._exit(3)
    from commander_amd import synth, healpix, shard
    from commander_amd.cr import build_context

    cfg = args.config
    nside = args.nside or synth.CONFIGS[cfg]["nside"]
    lmax = args.lmax or synth.CONFIGS[cfg]["lmax"]
    nband = len(synth.CONFIGS[cfg]["nu"])
    rings = pixels = bands = None
    lay = shard.rank_layout(nband, world, rank)
    if os.environ.get("CMDR_BENCH_SHARD"):          # "BxR": force band_parts x ring_parts
        bp, rp = (int(v) for v in os.environ["CMDR_BENCH_SHARD"].split("x"))
        lay = shard.rank_layout(nband, world, rank, bp, rp)
    ring_groups = None
    if world > 1:
        # band x ring-set hybrid: this rank owns lay["bands"] on ring set lay["ring_index"]
        if lay["ring_parts"] > 1:
            # ring ownership: blocks of 64 adjacent pairs dealt back and forth (healpix.rank_rings: the lanes of a
            # Legendre wave then hold neighbouring latitudes); CMDR_BENCH_RINGS=cyclic = Commander's own dealing
            rings = healpix.rank_rings(nside, lay["ring_index"], lay["ring_parts"],
                                       scheme=os.environ.get("CMDR_BENCH_RINGS", "block"))
            pixels = healpix.local_pixels(nside, rings)
        if lay["band_parts"] > 1:
            bands = lay["bands"]
            if lay["ring_parts"] > 1 and collective != "rccl-native":   # every rank creates every group, same order
                ring_groups = [dist.new_group([bg * lay["ring_parts"] + i for i in range(lay["ring_parts"])])
                               for bg in range(lay["band_parts"])]
    spec = synth.make_problem(cfg, nside=nside, lmax=lmax, pixels=pixels, bands=bands)
    emul = None
    if rehearse:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import emul_lib
        emul = emul_lib()
    ctx = build_context(spec, device=local_rank, rings_by_nside={nside: rings} if rings is not None else None,
                        _lib=emul)
    rccl_world = 0
    if dist is not None:
        import torch
        dev = "cuda:%d" % local_rank
        if collective == "rccl-native":
            # RCCL inside the library: ncclUniqueId from rank 0 through torch.distributed, then every sum over ranks is
            # an ncclAllReduce the library enqueues on its own stream (include/cmdr_hip.h, cmdr_ctx_init_rccl).  If the
            # run-time binding fails on ANY rank (librccl not loadable, communicator refused) all ranks switch together
            # to the same collective through torch.distributed on the library stream, and the JSON line says so.
            ok = 1
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(ctx.rccl_unique_id()), dtype=torch.uint8))
            except Exception as e:
                sys.stderr.write("bench.py rank %d: native RCCL unavailable (%r)\n" % (rank, e))
                ok = 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                dist.broadcast(idt, src=0)
                try:
                    ctx.init_rccl(bytes(idt.cpu().numpy().tobytes()), rank, world)
                    if lay["band_parts"] > 1:
                        ctx.rccl_split_rings(rank // lay["ring_parts"], lay["ring_index"], lay["ring_parts"])
                    rccl_world = ctx.rccl_size()
                except Exception as e:
                    sys.stderr.write("bench.py rank %d: native RCCL init failed (%r)\n" % (rank, e))
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                ctx.drop_rccl()      # a rank whose own init succeeded must not keep using its half-built communicators
                rccl_world = 0
                collective = "torch-stream (native RCCL binding failed, see stderr)"
                if lay["band_parts"] > 1 and lay["ring_parts"] > 1:
                    ring_groups = [dist.new_group([bg * lay["ring_parts"] + i for i in range(lay["ring_parts"])])
                                   for bg in range(lay["band_parts"])]
            else:
                assert rccl_world == world, (rccl_world, world)
        if collective != "rccl-native":
            views = {}

            def host_view(ptr, n):   # rehearsal: "device" memory of the emulation is host memory
                return torch.from_numpy(np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)),
                                                              shape=(n,)))

            def allreduce(ptr, n, stream):
                # RCCL all-reduce ordered on the library's own stream through torch: torch makes its NCCL stream wait
                # for the current stream before the collective and the current stream wait for it afterwards
                key = (ptr, n)
                if key not in views:
                    views[key] = torch.as_tensor(CudaView(ptr, n), device=dev)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                    dist.all_reduce(views[key])
            if world > 1 and lay["band_parts"] > 1:
                grp = ring_groups[rank // lay["ring_parts"]] if ring_groups else None

                def allreduce_rings(ptr, n):   # setup-time only (noise a_lm of a band over its ring sets)
                    if rehearse:
                        dist.all_reduce(host_view(ptr, n), group=grp)
                        return
                    t = torch.as_tensor(CudaView(ptr, n), device=dev)
                    if one_gpu:
                        h = t.cpu()
                        dist.all_reduce(h, group=grp)
                        t.copy_(h)
                    else:
                        dist.all_reduce(t, group=grp)
                    torch.cuda.synchronize()
                ctx.set_band_sharding(allreduce_rings if grp is not None else None, lay["ring_parts"])
            if collective == "blocking":   # the MPI-style blocking callback
                def allreduce_blocking(ptr, n):
                    if rehearse:
                        dist.all_reduce(host_view(ptr, n))
                        return
                    t = torch.as_tensor(CudaView(ptr, n), device=dev)
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
            rccl_world = dist.get_world_size() if not one_gpu else 0
    ctx.initPrecond()
    ctx.update_precond()
    resid, xi, eta = synth.draw_inputs(spec)
    dres = [ctx.dev(r.size, r) for r in resid]
    dxi = [ctx.dev(r.size, r) for r in xi]
    deta = ctx.dev(ctx.ncr, eta)
    b, x = ctx.dev(ctx.ncr), ctx.dev(ctx.ncr)
    L = ctx.L

    def step(refresh=False):
        if refresh:                      # the C_l-sampling chain refreshes the preconditioner every sample
            ctx.update_precond()         # (update_precond, comm_cr_mod.f90:76 -> updateDiffPrecond_diagonal)
        ctx.cr_computeRHS_dev("sample", dres, dxi, deta, None, b)
        return ctx.solve_dev(b, x, "fixed_iter", 1e-8, 5, NITER, 1)

    def barrier():
        L.cmdr_device_synchronize()
        if dist is not None:
            import torch
            if one_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
            if not rehearse:
                torch.cuda.synchronize()

    def timed(nsteps, refresh=False):
        barrier()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            out = step(refresh)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            import torch
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if one_gpu else "cuda:%d" % local_rank)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, out

    for _ in range(args.warmup):
        step()
    L.cmdr_profile_enable(ctx._h, 1)
    dt, (niter, stat, res) = timed(args.steps)
    ms = (ctypes.c_double * 6)()
    cnt = (ctypes.c_longlong * 6)()
    L.cmdr_profile_read_ext(ctx._h, 6, ms, cnt)
    L.cmdr_profile_enable(ctx._h, 0)
    info = (ctypes.c_int64 * 3)()
    L.cmdr_problem_info(ctx._h, info)
    dt_refresh = None
    if not args.no_extras:
        dt_refresh, _ = timed(max(1, min(args.steps, 3)), refresh=True)
        dt_refresh /= max(1, min(args.steps, 3))
    if rank == 0:
        nbm, steps_pruned = int(info[0]), int(info[2])
        avg = lambda k: ms[k] / max(int(cnt[k]), 1)   # noqa: E731  ms per launch span
        t_syn, t_ring, t_adj, t_mv = avg(0), avg(1), avg(2), avg(3)
        # ---- roofline of the dominant kernel.  With >= 6 maps per plan the Legendre adjoint of up to 9 maps is ONE launch
        # of k_leg_adj_mx (v_mfma_f64_16x16x4; profile kind 4); otherwise the VALU kernel k_leg_adj (kind 5).
        # F_SHT = 8 flop x (2 Nside ring pairs) x (lmax+1)(lmax+2)/2 (l, m) per scalar map
        # (2 FMA recursion + 2 FMA accumulate per (ring pair, l, m)); a launch = the Legendre stage of `nk` maps.
        npair_loc = (len(rings) if rings is not None else 2 * nside)
        use_mx = int(cnt[4]) > 0
        # nine maps: the ninth rides along in the same launch on the VALU (k_leg_adj_mx<.., X9>; CMDR_ADJ_X9=0: own launch)
        x9 = use_mx and nbm == 9 and os.environ.get("CMDR_ADJ_X9", "1") != "0"
        nk = (9 if x9 else min(nbm, 8)) if use_mx else nbm
        t_dom = avg(4) if use_mx else avg(5)
        f_alg = 8.0 * npair_loc * (lmax + 1) * (lmax + 2) / 2.0 * nk
        f_pruned = 8.0 * steps_pruned * nk                      # only (m, ring) inside libsharp's mlim cut are run
        if use_mx:    # executed: per (ring pair, l) one recursion step (mul + FMA = 3 flop) + 16 columns x 2 flop on the MFMA
            f_exec = (3.0 + 32.0 + (4.0 if x9 else 0.0)) * steps_pruned    # + 2 FMA per step for the map riding along
        else:
            nb_adj = 3.0 if nbm >= 3 else float(nbm)            # maps sharing one recursion in k_leg_adj<4,3>
            f_exec = (2.0 * 2.0 * nbm + 3.0 * nbm / nb_adj) * steps_pruned
        ach = f_alg / (t_dom * 1e-3) / 1e12 if t_dom else 0.0
        f_alg9 = f_alg / nk * nbm                               # all nbm maps (synthesis span, whole-matvec views)
        f_pruned9 = f_pruned / nk * nbm
        # ring stage, HBM view: reads 32 B + writes 32 B per (pair, m) phase entry + 8 B per pixel of the multiplier
        npix_loc = sum(s[0] for s in ctx.band_shape)
        b_ring = 64.0 * npair_loc * (lmax + 1) * nbm + 8.0 * npix_loc
        # whole-iteration HBM view (SURVEY.md 8d B_iter): per band 8(2 nalm + 3 npix) + 10*8*ncr
        b_iter = 8.0 * (2 * (lmax + 1) ** 2 * len(spec["bands"]) + 3 * npix_loc) + 80.0 * ctx.ncr
        traffic = None   # HBM bytes per adjoint launch from separate rocprofv3 --pmc passes (profiles/), if present
        tj = profile_summary("r02_pmc_traffic.json") or profile_summary("r01_pmc_traffic.json")
        if tj and world == 1 and cfg == "cfg3" and args.nside is None:
            traffic = (tj.get("adjoint_launch_bytes") or tj.get("legendre_span_bytes") or {}).get("mean")
        par = "single GPU"
        if world > 1:
            par = ("%d band groups x %d ring sets (band x ring-set hybrid)" % (lay["band_parts"], lay["ring_parts"])
                   if lay["band_parts"] > 1 else "ring-pair sharding x%d" % world)
            par += ", replicated a_lm, 1 all-reduce(ncr) per matvec"
        out = {
            "metric": "cg_solves_per_sec", "value": args.steps / dt, "unit": "solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic" if not rehearse else "REHEARSAL on the host emulation: not a measurement",
            "config": {"workload": "BASELINE.json configs[2]: %d Planck-like bands, CMB T-only, Nside=%d lmax=%d, "
                                   "amp-sample = cr_computeRHS + %d fixed PCG iterations, diagonal preconditioner"
                                   % (nband, nside, lmax, NITER),
                       "parallelism": par, "band_parts": lay["band_parts"], "ring_parts": lay["ring_parts"],
                       "ring_ownership": (os.environ.get("CMDR_BENCH_RINGS", "block") if lay["ring_parts"] > 1 else None),
                       "ncr": ctx.ncr, "cg_iters_per_sec": args.steps * NITER / dt},
            "rccl_world_size": rccl_world, "collective": collective if dist is not None else None,
            "value_with_precond_refresh": (1.0 / dt_refresh) if dt_refresh else None,
            "roofline": {
                "bound": "mfma",
                "bound_detail": ("fp64 matrix unit (v_mfma_f64_16x16x4_f64) + fp64 VALU recursion; one shared datapath on "
                                 "gfx950, fp64 matrix peak == fp64 vector peak") if use_mx else
                                "fp64 VALU (v_fma_f64); MI355X fp64 matrix peak == fp64 vector peak, so the same roof applies",
                "kernel": ("k_leg_adj_mx (Legendre adjoint on the matrix unit: phases -> a_lm, %d maps per launch)" % nk)
                          if use_mx else ("k_leg_adj (Legendre adjoint, VALU: phases -> a_lm, %d maps per span)" % nk),
                "achieved": ach, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS,
                "note": "achieved = SURVEY 8d algorithmic flops (8 per (ring pair, l, m) and map, unpruned) / launch time, as "
                        "the contract defines it; the kernel runs fewer: the (m, ring) cut removes 22 % of the steps and all maps "
                        "of a launch share one recursion, so frac can exceed 1.  frac_executed counts what the kernel really executes.",
                "traffic": traffic, "avg_launch_ms": t_dom, "launches": int(cnt[4] if use_mx else cnt[5]),
                "flop_per_launch": {"algorithmic_8d": f_alg, "mlim_pruned": f_pruned, "executed": f_exec},
                "frac_mlim_pruned": f_pruned / (t_dom * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_dom else None,
                "frac_executed": f_exec / (t_dom * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_dom else None,
                "secondary": {
                    "k_leg_adj_valu_leftover": {"bound": "fp64 valu", "avg_span_ms": avg(5) if use_mx and nbm > nk else None,
                                                "maps": nbm - nk if use_mx else 0},
                    "k_leg_synth": {"bound": "fp64 valu", "avg_span_ms": t_syn,
                                    "frac_algorithmic_8d": f_alg9 / (t_syn * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_syn else None,
                                    "frac_mlim_pruned": f_pruned9 / (t_syn * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_syn else None,
                                    "note": "8d charges the recursion to every map; the kernel shares it across up to "
                                            "5 maps, so the algorithmic fraction can exceed 1"},
                    "k_ring_fused": {"bound": "hbm", "avg_span_ms": t_ring, "algorithmic_bytes": b_ring,
                                     "achieved_GBs": b_ring / (t_ring * 1e-3) / 1e9 if t_ring else None,
                                     "frac": b_ring / (t_ring * 1e-3) / 1e9 / HBM_PEAK_GBS if t_ring else None},
                    "matvec": {"avg_ms": t_mv, "adjoint_span_ms": t_adj, "B_iter_bytes": b_iter,
                               "hbm_frac": b_iter / (t_mv * 1e-3) / 1e9 / HBM_PEAK_GBS if t_mv else None,
                               "fp64_frac_mlim_pruned": 2.0 * f_pruned9 / (t_mv * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_mv else None}}},
            "solve": {"niter": niter, "stat": stat, "res": res[0], "delta0": res[1]},
        }
        # second half of the headline metric: SHT pairs/s/GPU, the unit of commander3/src/sharp_test.f90:65-71
        # (one Yt followed by one Y), scalar and polarised, at the benchmark geometry, data resident in HBM
        if world == 1 and not args.no_extras and not rehearse:
            try:
                out["sht_pairs_per_sec_per_gpu"] = sht_pairs(L, nside, lmax)
                if args.nside is None:   # the reference's own SHT benchmark geometry, sharp_test.f90:31-33
                    out["sht_pairs_per_sec_per_gpu"]["sharp_test"] = sht_pairs(L, 2048, 3 * 2048, reps=5, pols=(False,))
            except Exception as e:
                out["sht_pairs_per_sec_per_gpu"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not rehearse:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, nside, lmax)
            except Exception as e:  # the baseline leg must never sink the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
