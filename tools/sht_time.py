"""Quick SHT timing on the GPU (development aid): pairs/s at a given geometry."""
import ctypes, sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import importlib
_m = importlib.import_module("commander_amd.lib"); lib, check = _m.lib, _m.check
from commander_amd import ShtPlan

nside = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
nmaps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
L = lib()
t0 = time.time()
plan = ShtPlan(nside, lmax, max_maps=nmaps)
print("plan build %.2fs" % (time.time() - t0), flush=True)
na, npx = plan.nalm, plan.npix
da, dm = ctypes.c_void_p(), ctypes.c_void_p()
check(L.cmdr_dev_alloc(na * 8 * nmaps, ctypes.byref(da)))
check(L.cmdr_dev_alloc(npx * 8 * nmaps, ctypes.byref(dm)))
rng = np.random.default_rng(0)
a = rng.standard_normal(na * nmaps)
check(L.cmdr_memcpy_h2d(da, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))
for job, name in [(1, "Y"), (2, "Yt")]:
    check(L.cmdr_sht_execute_dev(plan._h, job, nmaps, da, na, dm, npx))
    t = time.time()
    n = 5
    for _ in range(n):
        check(L.cmdr_sht_execute_dev(plan._h, job, nmaps, da, na, dm, npx))
    dt = (time.time() - t) / n
    flops = 8 * 2 * nside * (lmax + 1) * (lmax + 2) / 2 * nmaps
    print("%s nside=%d lmax=%d nmaps=%d: %.3f ms  -> %.2f TFLOP/s (algorithmic fp64)" % (name, nside, lmax, nmaps, dt * 1e3, flops / dt / 1e12), flush=True)
