"""Development timing of the CR path on the GPU: matvec, RHS, 40-iteration solve at a synth config."""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from commander_amd import synth
from commander_amd.cr import build_context

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
precond = sys.argv[2] if len(sys.argv) > 2 else "diagonal"
pol = "pol" in sys.argv[3:]
t0 = time.time(); spec = synth.make_problem(cfg, pol=True) if pol else synth.make_problem(cfg)
if "compact" in sys.argv[3:]:      # + monopole/dipole templates per band and 50 point sources
    synth.add_compact_blocks(spec, nsrc=50)
print("spec %.1fs" % (time.time() - t0), flush=True)
t0 = time.time(); ctx = build_context(spec); print("context %.1fs ncr=%d" % (time.time() - t0, ctx.ncr), flush=True)
t0 = time.time(); ctx.initPrecond(precond); print("initPrecond %.2fs" % (time.time() - t0), flush=True)
t0 = time.time(); ctx.update_precond(); print("update_precond %.2fs" % (time.time() - t0), flush=True)
resid, xi, eta = synth.draw_inputs(spec)
dres = [ctx.dev(r.size, r) for r in resid]; dxi = [ctx.dev(r.size, r) for r in xi]; deta = ctx.dev(ctx.ncr, eta)
b, x, y = ctx.dev(ctx.ncr), ctx.dev(ctx.ncr), ctx.dev(ctx.ncr)
import ctypes
ctx.L.cmdr_profile_enable(ctx._h, 1)
for rep in range(2):
    t0 = time.time(); ctx.cr_computeRHS_dev("sample", dres, dxi, deta, None, b); t1 = time.time() - t0
    t0 = time.time(); ctx.cr_matmulA_dev(b, y); t2 = time.time() - t0
    t0 = time.time(); n, stat, res = ctx.solve_dev(b, x, "fixed_iter", 1e-8, 5, 40, 1); t3 = time.time() - t0
    print("rhs %.1f ms  matvec %.1f ms  solve(40) %.1f ms  -> %.2f solves/s  (res %.3e / %.3e)" % (
        t1 * 1e3, t2 * 1e3, t3 * 1e3, 1.0 / (t1 + t3), res[0], res[1]), flush=True)
ms = (ctypes.c_double * 4)(); cnt = (ctypes.c_longlong * 4)()
ctx.L.cmdr_profile_read(ctx._h, ms, cnt)
print("per launch ms: synth %.3f ring %.3f adj %.3f matvec %.3f" % tuple(ms[k] / max(cnt[k], 1) for k in range(4)), flush=True)
