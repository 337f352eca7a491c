"""Worker process of tests/test_rccl_gpu.py: one RCCL scenario on cuda:0 in a FRESH process, libraries loaded in the
order bench.py loads them (torch first when torch is involved, so its bundled ROCm runtime is the one both use).
Prints "RCCL_WORKER_OK <mode>" on success."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1]
if mode in ("torch-stream", "native-after-torch"):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
import numpy as np  # noqa: E402
from commander_amd import synth  # noqa: E402
from commander_amd.cr import build_context  # noqa: E402

spec = synth.make_problem("cfg2", nside=64, lmax=128)
resid, xi, eta = synth.draw_inputs(spec)
x = np.random.default_rng(3).standard_normal(2 * 129 * 129)


def results(ctx):
    ctx.initPrecond()
    ctx.update_precond()
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    return [ctx.cr_matmulA(x), ctx.cr_invM(x), b, ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=10)[0]]


ref = results(build_context(spec))
ctx = build_context(spec)
if mode == "sliced":       # one scalar component (the benchmark's shape): sliced CG vectors through ncclReduceScatter / ncclAllGather
    spec = synth.make_problem("cfg3", nside=64, lmax=128)
    resid, xi, eta = synth.draw_inputs(spec)
    x = np.random.default_rng(3).standard_normal(129 * 129)
    ref = results(build_context(spec))
    ctx = build_context(spec)
    os.environ["CMDR_SLICE_FORCE"] = "1"
if mode in ("native", "native-after-torch", "split", "sliced"):
    assert ctx.L.cmdr_rccl_version() >= 20000, ctx.L.cmdr_last_error()
    ctx.init_rccl(ctx.rccl_unique_id(), 0, 1)
    assert ctx.rccl_size() == 1
    if mode == "split":
        ctx.rccl_split_rings(0, 0, 1)      # band group 0, ring set 0 of 1: the hybrid bookkeeping with trivial groups
    if mode == "sliced":
        ctx.set_vector_slicing(0, 1)
elif mode == "torch-stream":
    from bench import CudaView
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    calls = []

    def allreduce(ptr, n, stream):
        t = torch.as_tensor(CudaView(ptr, n), device="cuda:0")
        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device="cuda:0")):
            dist.all_reduce(t)
        calls.append(n)
    ctx.set_allreduce_stream(allreduce)
else:
    raise SystemExit("unknown mode " + mode)
got = results(ctx)
for a, b in zip(got, ref):
    assert np.array_equal(a, b), float(np.abs(a - b).max())
if mode == "torch-stream":
    # the sums over ranks go out as row ranges of the stacked vector (two halves, the first on the second stream) or whole
    assert calls and max(calls) <= ctx.ncr and sum(calls) >= ctx.ncr, (len(calls), max(calls), ctx.ncr)
    ctx.close()
    dist.destroy_process_group()
print("RCCL_WORKER_OK", mode, flush=True)
