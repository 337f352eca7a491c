"""Development timing of ONE rank's share of a ring-sharded matvec (no collective): python tools/cr_time_rank.py cfg3 8
times what rank 0 of 8 would compute per iteration, to tune the small-shard kernel parameters on a 1-GPU box."""
import sys, time, ctypes
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from commander_amd import synth, healpix
from commander_amd.cr import build_context

from commander_amd import shard
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nside = synth.CONFIGS[cfg]["nside"]
nband = len(synth.CONFIGS[cfg]["nu"])
if len(sys.argv) > 3:                       # "BxR": band groups x ring sets; time the most loaded rank
    bp, rp = (int(v) for v in sys.argv[3].split("x"))
else:
    bp, rp = shard.plan_shards(nband, world)
lay = max((shard.rank_layout(nband, world, r, bp, rp) for r in range(world)), key=lambda l: len(l["bands"]))
scheme = os.environ.get("CMDR_BENCH_RINGS", "block")
rings = healpix.rank_rings(nside, lay["ring_index"], rp, scheme=scheme) if rp > 1 else None
pix = healpix.local_pixels(nside, rings) if rings is not None else None
spec = synth.make_problem(cfg, pixels=pix, bands=lay["bands"] if bp > 1 else None)
ctx = build_context(spec, rings_by_nside={nside: rings} if rings is not None else None)
print("layout: %d band groups x %d ring sets (%s ring ownership); this rank: %d bands" % (bp, rp, scheme, len(lay["bands"])), flush=True)
ctx.initPrecond(); ctx.update_precond()
x, y, b = ctx.dev(ctx.ncr, np.random.default_rng(0).standard_normal(ctx.ncr)), ctx.dev(ctx.ncr), ctx.dev(ctx.ncr)
ctx.L.cmdr_profile_enable(ctx._h, 1)
for rep in range(3):
    ctx.cr_matmulA_dev(x, y)
ctx.sync() if hasattr(ctx, "sync") else None
t0 = time.time()
n = 20
for _ in range(n):
    ctx.cr_matmulA_dev(x, y)
    ctx.cr_invM_dev(y, b)
ctx.L.cmdr_memcpy_d2h(np.zeros(1).ctypes.data_as(ctypes.c_void_p), y.ptr, 8)
dt = (time.time() - t0) / n
ms = (ctypes.c_double * 6)(); cnt = (ctypes.c_longlong * 6)()
ctx.L.cmdr_profile_read_ext(ctx._h, 6, ms, cnt)
print("world=%d rank share: matvec+invM %.3f ms | per launch ms: synth %.3f ring %.3f adj %.3f (matrix-unit %.3f + VALU %.3f) matvec %.3f" % (
    (world, dt * 1e3) + tuple(ms[k] / max(cnt[k], 1) for k in (0, 1, 2, 4, 5, 3))), flush=True)
