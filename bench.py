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
    """Reported, not targeted: the build's own CPU oracle (oracle/: numpy + C/OpenMP restatement,
    SIMD-blocked Legendre stage, the same algorithm class as libsharp2) timed on the host cores this job is granted.
    It is NOT the Fortran + libsharp2 binary (unbuildable here: no HEALPix / libsharp2 / FFTW / gfortran).
      * benchmark configuration: ONE real cr_matmulA on all nine bands (t_matvec) and t_sht_pair; `value` = 1 / (41.5
        matvec-equivalents): RHS ~ half a matvec + 41 matvecs (r = b - A x0 is skipped for x0 = 0; M^-1 is negligible)
      * configs[1] (3 bands, CMB + synch, Nside 256 / lmax 512): t_sht_pair, t_matvec (medians) and a REAL t_solve
        (cr_computeRHS + preconditioner refresh + 40 fixed PCG iterations)"""
    import numpy as np
    from commander_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_system
    from oracle import sht as osht
    cores = host_cores()
    os.environ["ORACLE_THREADS"] = str(cores)
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def med(f, n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def sht_pair_times(ns, lm, n):
        rng = np.random.default_rng(1)
        m = rng.standard_normal(12 * ns * ns)
        q, u = rng.standard_normal(12 * ns * ns), rng.standard_normal(12 * ns * ns)
        osht.Y(ns, lm, osht.Yt(ns, lm, m))     # plans / twiddles
        t0 = med(lambda: osht.Y(ns, lm, osht.Yt(ns, lm, m)), n)
        t2 = med(lambda: osht.sht_spin2(1, ns, lm, *[None] * 0, **dict(zip(("almE", "almB"), osht.sht_spin2(2, ns, lm, mapQ=q, mapU=u)))), n)
        return t0, t2
    # ---- configs[1]: everything measured for real
    c2 = synth.CONFIGS["cfg2"]
    spec2 = synth.make_problem("cfg2")
    S2 = oracle_system(spec2)
    S2.init_precond_diag()
    S2.update_precond_diag()
    x2 = np.random.default_rng(0).standard_normal(S2.ncr)
    S2.matmulA(x2)
    t_mv2 = med(lambda: S2.matmulA(x2), 5)
    resid, xi, eta = synth.draw_inputs(spec2)
    col = lambda lst: [np.asarray(v).reshape(len(v), -1) for v in lst]   # noqa: E731
    t0 = time.perf_counter()
    b2 = S2.computeRHS(col(resid), "sample", col(xi), eta)
    S2.update_precond_diag()
    S2.solve(b2, "fixed_iter", 1e-8, 5, NITER, 1)
    t_solve2 = time.perf_counter() - t0
    p0_2, p2_2 = sht_pair_times(c2["nside"], c2["lmax"], 5)
    # ---- the benchmark configuration: one real matvec on all bands
    spec = synth.make_problem(cfg, nside=nside, lmax=lmax)
    S = oracle_system(spec)
    x = np.random.default_rng(0).standard_normal(S.ncr)
    t0 = time.perf_counter()
    S.matmulA(x)
    t_mv = time.perf_counter() - t0
    p0, p2 = sht_pair_times(nside, lmax, 1)
    mv_per_solve = NITER + 1 + 0.5
    return {"value": 1.0 / (mv_per_solve * t_mv), "unit": "solves/s", "cores": cores, "kind": "port", "cpu": cpu,
            "matvec_s": t_mv, "sht_pair_s": {"spin0": p0, "spin2_QU": p2},
            "sample": "one oracle cr_matmulA (oracle/cr_oracle.py + oracle/sht_oracle.c, SIMD-blocked Legendre stage, OpenMP "
                      "x%d) on all %d bands at Nside=%d lmax=%d: %.1f s; x %.1f matvec-equivalents per solve.  The CPU "
                      "restatement, NOT the Fortran + libsharp2 binary"
                      % (cores, len(spec["bands"]), nside, lmax, t_mv, mv_per_solve),
            "configs1": {"config": "configs[1]: 3 bands, CMB+synch, Nside %d lmax %d" % (c2["nside"], c2["lmax"]),
                         "t_sht_pair_s": {"spin0": p0_2, "spin2_QU": p2_2}, "t_matvec_s": t_mv2, "t_solve_s": t_solve2,
                         "solves_per_sec": 1.0 / t_solve2,
                         "note": "t_solve = cr_computeRHS + preconditioner refresh + 40 fixed PCG iterations, measured"}}


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


# ------------------------------------------------------------------------------------------------ executed-flop model
def _batches(n, nbmax):
    """launch_leg_synth / launch_leg_adj's balanced split of n maps into batches of at most nbmax (kernels.hip for_batches)."""
    nb = (n + nbmax - 1) // nbmax
    out, k0 = [], 0
    for ib in range(nb):
        sz = (n - k0 + (nb - ib) - 1) // (nb - ib)
        out.append(sz)
        k0 += sz
    return out


def _batches_left(nT, env=os.environ):
    """scalar maps the matrix-unit adjoint launches leave to the DPP / VALU form"""
    mx_min = int(env.get("CMDR_ADJ_MX", "6"))
    left = nT
    while mx_min > 0 and left >= mx_min:
        nb = min(8, left)
        x9 = env.get("CMDR_ADJ_X9", "1") != "0" and nb == 8 and left - 8 == 1
        left -= nb + (1 if x9 else 0)
    return left


def executed_flops(nT, npol, steps0, steps2, env=os.environ):
    """fp64 operations the Legendre launches of ONE matvec really execute, per launch kind (what `roofline.frac` is
    priced on; the SURVEY 8d convention -- 8 flop per (ring pair, l, m) and scalar map, 24 per (Q,U) pair, every
    (m, ring) combination -- is kept beside it as `algorithmic_8d`).  Per useful (ring pair, l) recursion step:
      scalar synthesis (k_leg_synth_wg, batches of <= 5 maps sharing one recursion): 3 + 4 nb
      scalar adjoint: matrix unit (6..9 maps: 3 + 32 on 16 MFMA columns, + 4 for a ninth map on the VALU),
                      DPP form (3..5 maps: 3 + 4 nb), VALU form (1..2 maps: 3 + 4 nb + the wave-wide reduction, not counted)
      spin-2 synthesis: 8 (two chains) + 2 (W, X) + 16 per (Q,U) pair; four / three / two pairs share the chains
      spin-2 adjoint:   matrix unit (3..4 pairs per launch): 8 + 64 (two A operands x 16 columns);
                        VALU: 8 + 2 + 16 per pair (two pairs share the chains), reduction not counted"""
    f = {"synth0": 0.0, "adj_mx": 0.0, "adj_valu": 0.0, "synth2": 0.0, "adj2": 0.0, "adj2_kernel": None}
    for nb in _batches(nT, 5) if nT else []:
        f["synth0"] += (3.0 + 4.0 * nb) * steps0
    mx_min = int(env.get("CMDR_ADJ_MX", "6"))
    left = nT
    while mx_min > 0 and left >= mx_min:
        nb = min(8, left)
        x9 = env.get("CMDR_ADJ_X9", "1") != "0" and nb == 8 and left - 8 == 1
        f["adj_mx"] += (3.0 + 32.0 + (4.0 if x9 else 0.0)) * steps0
        left -= nb + (1 if x9 else 0)
    if left >= 3 and env.get("CMDR_ADJ_DX", "1") != "0":
        f["adj_valu"] += (3.0 + 4.0 * left) * steps0
    elif left:
        for nb in _batches(left, 3):
            f["adj_valu"] += (3.0 + 4.0 * nb) * steps0
    if npol:
        p = npol
        f["synth2_launches"] = 0
        while p >= 3 and env.get("CMDR_SYNTH2_NP", "1") != "0":      # k_leg2_synth_npx: 4 / 3 pairs share the chains
            nb = 3 if p in (3, 5, 6) else 4
            f["synth2"] += (10.0 + 16.0 * nb) * steps2
            f["synth2_launches"] += 1
            p -= nb
        while p >= 2:
            f["synth2"] += (10.0 + 32.0) * steps2
            f["synth2_launches"] += 1
            p -= 2
        if p:
            f["synth2"] += 26.0 * steps2
            f["synth2_launches"] += 1
        mx2 = int(env.get("CMDR_ADJ2_MX", "3"))
        p = npol
        names = []
        while mx2 > 0 and p >= mx2:
            f["adj2"] += (8.0 + 64.0) * steps2
            p -= min(4, p)
            names.append("k_leg2_adj_mx")
        while p >= 2:
            f["adj2"] += (10.0 + 32.0) * steps2
            p -= 2
            names.append("k_leg2_adj_np2")
        if p:
            f["adj2"] += 26.0 * steps2
            names.append("k_leg2_adj")
        f["adj2_launches"] = len(names)
        f["adj2_kernel"] = "+".join(sorted(set(names)))
    return f


def kernel_lines(ms, cnt, nT, npol, steps0, steps2):
    """Per launch kind: average ms per matvec-span and the executed fp64 rate.  Kinds (cmdr_profile_read_ext): 0 all
    synthesis launches, 2 all adjoint launches, 4 matrix-unit scalar adjoint, 5 other scalar adjoint, 6 / 7 spin-2."""
    avg = lambda k: ms[k] / max(int(cnt[k]), 1)   # noqa: E731
    f = executed_flops(nT, npol, steps0, steps2)
    t = {"synth0": avg(0) - (avg(6) if int(cnt[6]) else 0.0), "adj_mx": avg(4) if int(cnt[4]) else 0.0,
         "adj_valu": avg(5) if int(cnt[5]) else 0.0, "synth2": avg(6) if int(cnt[6]) else 0.0,
         "adj2": avg(7) if int(cnt[7]) else 0.0}
    name = {"synth0": "k_leg_synth_wg (scalar Legendre synthesis, all batches of one matvec)",
            "adj_mx": "k_leg_adj_mx (scalar Legendre adjoint on the matrix unit)",
            "adj_valu": "k_leg_adj_dx / k_leg_adj (scalar Legendre adjoint, DPP / VALU form)",
            "synth2": "k_leg2_synth[_np2|_npx] (spin-2 Legendre synthesis, all (Q,U) pairs of one matvec)",
            "adj2": "%s (spin-2 Legendre adjoint, all (Q,U) pairs of one matvec)" % (f["adj2_kernel"] or "k_leg2_adj")}
    nl = {"synth0": len(_batches(nT, 5)) if nT else 0, "adj_mx": 1, "adj_valu": 1, "synth2": max(1, f.get("synth2_launches", 1)),
          "adj2": max(1, f.get("adj2_launches", 1))}
    out = {}
    for k in t:
        if t[k] > 0.0 and f[k] > 0.0:
            tf = f[k] / (t[k] * 1e-3) / 1e12
            out[k] = {"kernel": name[k], "avg_ms": t[k], "launches_per_matvec": nl[k], "avg_launch_ms": t[k] / max(nl[k], 1),
                      "executed_flop": f[k], "executed_tflops": tf, "frac_executed": tf / FP64_PEAK_TFLOPS}
    return out


def other_config(cfg, preconds=("diagonal",), pol=None, sht=None, label=None):
    """One more BASELINE.json configuration on this GPU (BASELINE.md section 3's table): build, then per preconditioner
    one warm-up sample and ONE timed sample (cr_computeRHS + 40 fixed PCG iterations, inputs resident) with the
    per-kernel executed fractions.  Returns one record per preconditioner."""
    from commander_amd import synth
    from commander_amd.cr import build_context
    t_build = time.perf_counter()
    spec = synth.make_problem(cfg) if pol is None else synth.make_problem(cfg, pol=pol)
    ctx = build_context(spec)
    L = ctx.L
    resid, xi, eta = synth.draw_inputs(spec)
    dres = [ctx.dev(r.size, r) for r in resid]
    dxi = [ctx.dev(r.size, r) for r in xi]
    deta = ctx.dev(ctx.ncr, eta)
    b, x = ctx.dev(ctx.ncr), ctx.dev(ctx.ncr)
    t_build = time.perf_counter() - t_build
    c = synth.CONFIGS[cfg]
    recs = []

    def step():
        ctx.cr_computeRHS_dev("sample", dres, dxi, deta, None, b)
        return ctx.solve_dev(b, x, "fixed_iter", 1e-8, 5, NITER, 1)
    for precond in preconds:
        t_pre = time.perf_counter()
        ctx.initPrecond(precond)
        ctx.update_precond()
        t_pre = time.perf_counter() - t_pre
        step()
        L.cmdr_device_synchronize()
        t0 = time.perf_counter()
        niter, stat, res = step()            # the timed sample: no event spans (fixed_iter solves replay a hipGraph)
        L.cmdr_device_synchronize()
        dt = time.perf_counter() - t0
        L.cmdr_profile_enable(ctx._h, 1)     # one more sample for the per-kernel breakdown
        step()
        L.cmdr_device_synchronize()
        ms = (ctypes.c_double * 8)()
        cnt = (ctypes.c_longlong * 8)()
        L.cmdr_profile_read_ext(ctx._h, 8, ms, cnt)
        L.cmdr_profile_enable(ctx._h, 0)
        info = (ctypes.c_int64 * 8)()
        L.cmdr_problem_info_ext(ctx._h, 8, info)
        nT, npol, steps0, steps2 = int(info[4]), int(info[5]), int(info[2]), int(info[3])
        kl = kernel_lines(ms, cnt, nT, npol, steps0, steps2)
        dom = max(kl.values(), key=lambda v: v["avg_launch_ms"]) if kl else None     # the longest single launch
        recs.append({"config": label or cfg, "nside": c["nside"], "lmax": c["lmax"], "bands": len(c["nu"]),
                     "components": list(c["comps"]), "polarised": bool(c.get("pol", False)) if pol is None else bool(pol),
                     "preconditioner": precond, "plans": int(info[7]), "ncr": ctx.ncr, "solves_per_sec": 1.0 / dt,
                     "ms_per_step": dt * 1e3, "niter": niter, "matvec_ms": ms[3] / max(int(cnt[3]), 1),
                     "dominant_kernel": dom, "kernels": kl, "setup_s": t_build, "precond_setup_s": t_pre})
    if sht:
        recs[0]["sht_pairs_per_sec_per_gpu"] = sht_pairs(L, c["nside"], c["lmax"], reps=3)
    for a in dres + dxi + [deta, b, x]:
        a.free()
    ctx.close()
    return recs


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
    # m-sliced CG vectors (reduce-scatter + all-gather instead of the all-reduce, 1 / N of the vector kernels): opt-in.  At
    # the benchmark's size the two extra 8 KB sums per iteration and the loss of the two-half overlap are estimated to cost
    # what the replicated vector kernels cost; it pays for larger ncr / more ranks.
    sliced = world > 1 and lay["band_parts"] == 1 and os.environ.get("CMDR_BENCH_SLICE") == "1"
    if sliced:
        ctx.set_vector_slicing(rank, world)
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
    NK = 8
    ms = (ctypes.c_double * NK)()
    cnt = (ctypes.c_longlong * NK)()
    L.cmdr_profile_read_ext(ctx._h, NK, ms, cnt)
    L.cmdr_profile_enable(ctx._h, 0)
    info = (ctypes.c_int64 * 8)()
    L.cmdr_problem_info_ext(ctx._h, 8, info)
    dt_refresh = None
    if not args.no_extras:
        dt_refresh, _ = timed(max(1, min(args.steps, 3)), refresh=True)
        dt_refresh /= max(1, min(args.steps, 3))
    # the same step through the HOST-pointer ABI (cmdr_compute_rhs + cmdr_solve: what INTEGRATION.md's Level-1 Fortran
    # driver calls): pays the H2D of the residual / noise maps and eta and the D2H of rhs and x every sample.  Never `value`.
    dt_host = dt_host_pinned = None
    if not args.no_extras and world == 1 and not rehearse:
        def host_step():
            bh = ctx.cr_computeRHS("sample", resid, xi, eta)
            return ctx.solve_cr_eqn_by_CG(bh, "fixed_iter", 1e-8, 5, NITER, 1)
        host_step()
        barrier()
        t0 = time.perf_counter()
        host_step()
        barrier()
        dt_host = time.perf_counter() - t0
        # ... and with the driver's long-lived input arrays page-locked once (cmdr_host_register), as INTEGRATION.md
        # recommends for data(i)%res%map and the noise draws
        dt_host_pinned = None
        try:
            pinned = [np.ascontiguousarray(a, dtype=np.float64) for a in list(resid) + list(xi) + [eta]]
            resid_p, xi_p, eta_p = pinned[:len(resid)], pinned[len(resid):len(resid) + len(xi)], pinned[-1]
            for a in pinned:
                ctx.host_register(a)
            try:
                def host_step_pinned():
                    bh = ctx.cr_computeRHS("sample", resid_p, xi_p, eta_p)
                    return ctx.solve_cr_eqn_by_CG(bh, "fixed_iter", 1e-8, 5, NITER, 1)
                host_step_pinned()
                barrier()
                t0 = time.perf_counter()
                host_step_pinned()
                barrier()
                dt_host_pinned = time.perf_counter() - t0
            finally:
                for a in pinned:
                    ctx.host_unregister(a)
        except Exception as e:   # noqa: BLE001  (reported leg only)
            print("[bench] pinned host-ABI leg skipped: %r" % (e,), file=sys.stderr, flush=True)
    if rank == 0:
        nbm, steps_pruned = int(info[0]), int(info[2])
        nT, npol, steps2 = int(info[4]), int(info[5]), int(info[3])
        avg = lambda k: ms[k] / max(int(cnt[k]), 1)   # noqa: E731  ms per launch span
        t_syn, t_ring, t_adj, t_mv = avg(0), avg(1), avg(2), avg(3)
        kl = kernel_lines(ms, cnt, nT, npol, steps_pruned, steps2)
        # ---- roofline of the dominant kernel = the longest single Legendre launch of a matvec.  Headline configuration:
        # k_leg_adj_mx, ONE launch = the Legendre adjoint of all nine maps on the fp64 matrix unit.
        npair_loc = (len(rings) if rings is not None else 2 * nside)
        dom_key = max(kl, key=lambda k: kl[k]["avg_launch_ms"]) if kl else None
        dom = kl[dom_key] if dom_key else {"kernel": None, "avg_ms": 0.0, "executed_flop": 0.0, "executed_tflops": 0.0,
                                           "frac_executed": 0.0, "launches_per_matvec": 1, "avg_launch_ms": 0.0}
        t_dom = dom["avg_ms"]                  # one launch for the matrix-unit adjoint; else the span of that kind's launches
        nk = npol if dom_key in ("synth2", "adj2") else nT
        if dom_key == "adj_mx":
            nk = nT - _batches_left(nT)
        elif dom_key == "adj_valu":
            nk = _batches_left(nT)
        # 8 flop x (ring pairs) x (lmax+1)(lmax+2)/2 per scalar map (24 per (Q,U) pair), every (m, ring)
        unit = 24.0 if dom_key in ("synth2", "adj2") else 8.0
        f_alg = unit * npair_loc * (lmax + 1) * (lmax + 2) / 2.0 * nk
        f_pruned = unit * (steps2 if dom_key in ("synth2", "adj2") else steps_pruned) * nk
        tfl = lambda fl, t: (fl / (t * 1e-3) / 1e12) if t else None   # noqa: E731
        f_alg9 = 8.0 * npair_loc * (lmax + 1) * (lmax + 2) / 2.0 * nT
        f_pruned9 = 8.0 * steps_pruned * nT
        # ring stage, HBM view: reads 32 B + writes 32 B per (pair, m) phase entry + 8 B per pixel of the multiplier
        npix_loc = sum(s[0] for s in ctx.band_shape)
        b_ring = 64.0 * npair_loc * (lmax + 1) * nbm + 8.0 * npix_loc
        # whole-iteration HBM view (SURVEY.md 8d B_iter): per band 8(2 nalm + 3 npix) + 10*8*ncr
        b_iter = 8.0 * (2 * (lmax + 1) ** 2 * len(spec["bands"]) + 3 * npix_loc) + 80.0 * ctx.ncr
        # HBM bytes per launch of the dominant kernel: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
        # command, collected with tools/collect_profiles.sh and STORED under profiles/ (not measured in this run)
        traffic, traffic_src = None, None
        if world == 1 and cfg == "cfg3" and args.nside is None and dom_key == "adj_mx":
            for name in ("r03_pmc_traffic.json", "r02_pmc_traffic.json"):
                tj = profile_summary(name)
                traffic = ((tj or {}).get("adjoint_launch_bytes") or {}).get("mean")
                if traffic:
                    traffic_src = "profiles/" + name
                    break
        par = "single GPU"
        if world > 1:
            par = ("%d band groups x %d ring sets (band x ring-set hybrid)" % (lay["band_parts"], lay["ring_parts"])
                   if lay["band_parts"] > 1 else "ring-pair sharding x%d" % world)
            par += (", m-sliced CG vectors: reduce-scatter(ncr) + all-gather(ncr) + two 8 KB all-reduces per iteration" if sliced
                    else ", replicated a_lm, the all-reduce(ncr) of a matvec in two halves, the first beside the second half's adjoint")
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
            "value_host_abi": (1.0 / dt_host) if dt_host else None,
            "value_host_abi_pinned": (1.0 / dt_host_pinned) if dt_host and dt_host_pinned else None,
            "value_host_abi_note": "the same step through the host-pointer ABI (cmdr_compute_rhs + cmdr_solve: H2D of the "
                                   "residual / noise maps and eta, D2H of rhs and x, every sample); _pinned: with those input arrays page-locked "
                                   "once through cmdr_host_register; reported, never `value`",
            "roofline": {
                "bound": "mfma",
                "bound_detail": "fp64 matrix unit (v_mfma_f64_16x16x4_f64) + fp64 VALU recursion; one shared datapath on "
                                "gfx950, fp64 matrix peak == fp64 vector peak (78.6 TFLOP/s)",
                "kernel": dom["kernel"], "maps_per_launch": nk,
                "achieved": dom["executed_tflops"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": dom["frac_executed"],
                "note": "achieved / frac = fp64 operations the kernel EXECUTES per launch (executed_flops() in bench.py: per "
                        "useful (ring pair, l) step 3 flop of recursion shared by all maps, 32 on the matrix unit, 4 for the "
                        "ninth map) / average launch time measured with HIP events on the library stream.  The SURVEY 8d "
                        "convention (8 flop per (ring pair, l, m) AND map, every (m, ring) combination) charges work the "
                        "kernel does not do -- one recursion serves all maps, 22 % of the (m, ring) steps lie outside "
                        "libsharp's mlim cut -- and is kept as algorithmic_8d / mlim_pruned (speed equivalents, may exceed 1)",
                "avg_launch_ms": t_dom, "launches": int(cnt[4] if dom_key == "adj_mx" else cnt[2]),
                "flop_per_launch": {"executed": dom["executed_flop"], "algorithmic_8d": f_alg, "mlim_pruned": f_pruned},
                "algorithmic_8d": {"tflops_equiv": tfl(f_alg, t_dom), "frac_equiv": (tfl(f_alg, t_dom) or 0.0) / FP64_PEAK_TFLOPS},
                "mlim_pruned": {"tflops_equiv": tfl(f_pruned, t_dom), "frac_equiv": (tfl(f_pruned, t_dom) or 0.0) / FP64_PEAK_TFLOPS},
                "traffic": traffic, "traffic_stored_profile": {"bytes_per_launch": traffic, "source": traffic_src,
                                                               "note": "FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc "
                                                                       "passes of this command (tools/collect_profiles.sh), "
                                                                       "stored under profiles/; NOT measured in this run"},
                "kernels": kl,
                "secondary": {
                    "k_ring_fused": {"bound": "hbm", "avg_span_ms": t_ring, "algorithmic_bytes": b_ring,
                                     "achieved_GBs": b_ring / (t_ring * 1e-3) / 1e9 if t_ring else None,
                                     "peak_GBs": HBM_PEAK_GBS,
                                     "frac": b_ring / (t_ring * 1e-3) / 1e9 / HBM_PEAK_GBS if t_ring else None},
                    "matvec": {"avg_ms": t_mv, "synth_span_ms": t_syn, "adjoint_span_ms": t_adj, "B_iter_bytes": b_iter,
                               "hbm_frac": b_iter / (t_mv * 1e-3) / 1e9 / HBM_PEAK_GBS if t_mv else None,
                               "fp64_frac_executed": (sum(v["executed_flop"] for v in kl.values()) / (t_mv * 1e-3) / 1e12
                                                      / FP64_PEAK_TFLOPS) if t_mv else None,
                               "fp64_frac_equiv_mlim_pruned": 2.0 * f_pruned9 / (t_mv * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_mv else None,
                               "fp64_frac_equiv_algorithmic_8d": 2.0 * f_alg9 / (t_mv * 1e-3) / 1e12 / FP64_PEAK_TFLOPS if t_mv else None}}},
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
        # the other BASELINE.json configurations on this GPU: one timed sample each
        if world == 1 and not args.no_extras and not rehearse and cfg == "cfg3" and args.nside is None:
            t0 = time.perf_counter()
            oc = []
            for a in (dict(cfg="cfg2", label="configs[1]: 3 bands, CMB+synch, Nside 256 lmax 512"),
                      dict(cfg="cfg4", sht=True, label="configs[3]: T/E/B CMB, one band, Nside 2048 lmax 4000 (SHT roofline run)"),
                      dict(cfg="cfg5", preconds=("diagonal", "pseudoinv"),
                           label="configs[4]: 5 diffuse components (2 with varying mixing), 9 bands, Nside 1024"),
                      dict(cfg="cfg3", pol=True, label="configs[2] geometry with T,Q,U bands and a T/E/B component")):
                try:
                    oc += other_config(**a)
                except Exception as e:
                    oc.append({"config": a.get("label"), "error": repr(e)})
            out["other_configs"] = oc
            out["other_configs_wall_s"] = time.perf_counter() - t0
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
