"""GPU tier: the RCCL paths of the multi-GPU design on the one GPU a test box has.

* the library's own RCCL binding (cmdr_ctx_init_rccl: dlopen'ed librccl, ncclAllReduce on the library stream), 1-rank
  communicator, incl. the ncclCommSplit of the band x ring-set hybrid: results bit-equal to the non-distributed context;
* the stream-ordered callback through torch.distributed ("nccl" == RCCL) with torch.cuda.ExternalStream;
* `bench.py --gpus 2` launching two ranks by itself (both on device 0, collectives through gloo on host copies: RCCL
  refuses two ranks on one device) and reporting n_gpus = 2.
What mpi_dot_product / libsharp2's exchange do in the reference (comm_utils.f90:599-614); SURVEY.md 8e."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def problem():
    from commander_amd import synth
    from commander_amd.cr import build_context
    spec = synth.make_problem("cfg2", nside=64, lmax=128)
    ctx = build_context(spec)
    ctx.initPrecond()
    ctx.update_precond()
    rng = np.random.default_rng(3)
    x = rng.standard_normal(ctx.ncr)
    resid, xi, eta = synth.draw_inputs(spec)
    b = ctx.cr_computeRHS("sample", resid, xi, eta)
    sol = ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=10)[0]
    return spec, x, ctx.cr_matmulA(x), ctx.cr_invM(x), b, sol


def _same(ctx, problem):
    from commander_amd import synth
    spec, x, y, pm, b, sol = problem
    ctx.initPrecond()
    ctx.update_precond()
    assert np.array_equal(ctx.cr_matmulA(x), y)
    assert np.array_equal(ctx.cr_invM(x), pm)
    resid, xi, eta = synth.draw_inputs(spec)
    assert np.array_equal(ctx.cr_computeRHS("sample", resid, xi, eta), b)
    assert np.array_equal(ctx.solve_cr_eqn_by_CG(b, "fixed_iter", maxiter=10)[0], sol)


def test_native_rccl_one_rank_bit_equal(problem):
    from commander_amd.cr import build_context
    ctx = build_context(problem[0])
    assert ctx.L.cmdr_rccl_version() >= 20000
    ctx.init_rccl(ctx.rccl_unique_id(), 0, 1)
    assert ctx.rccl_size() == 1
    _same(ctx, problem)


def test_native_rccl_split_rings_one_rank(problem):
    from commander_amd.cr import build_context
    ctx = build_context(problem[0])
    ctx.init_rccl(ctx.rccl_unique_id(), 0, 1)
    ctx.rccl_split_rings(0, 0, 1)          # band group 0, ring set 0 of 1: the hybrid bookkeeping with trivial groups
    _same(ctx, problem)


def test_torch_nccl_stream_ordered_callback_one_rank(problem):
    import torch
    import torch.distributed as dist
    from commander_amd.cr import build_context
    sys.path.insert(0, ROOT)
    from bench import CudaView
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        ctx = build_context(problem[0])
        calls = []

        def allreduce(ptr, n, stream):
            t = torch.as_tensor(CudaView(ptr, n), device="cuda:0")
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device="cuda:0")):
                dist.all_reduce(t)
            calls.append(n)
        ctx.set_allreduce_stream(allreduce)
        _same(ctx, problem)
        assert calls and max(calls) == ctx.ncr
    finally:
        dist.destroy_process_group()


def test_bench_gpus_2_launches_two_ranks():
    env = dict(os.environ, CMDR_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--config", "cfg2", "--nside", "64", "--lmax", "128", "--no-extras"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["config"]["ring_parts"] == 2
    assert lines[0]["data"] == "synthetic" and lines[0]["value"] > 0 and lines[0]["solve"]["niter"] == 40
