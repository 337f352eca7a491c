"""GPU tier: the RCCL paths of the multi-GPU design on the one GPU a test box has, each in a fresh process
(tests/rccl_worker.py) that loads its libraries in bench.py's order.

* the library's own RCCL binding (cmdr_ctx_init_rccl: dlopen'ed librccl, ncclAllReduce on the library stream), 1-rank
  communicator, alone and in a process that holds torch's bundled RCCL, incl. the ncclCommSplit of the band x ring-set
  hybrid: cr_matmulA / cr_invM / cr_computeRHS / a 10-iteration solve bit-equal to the non-distributed context;
* the stream-ordered callback through torch.distributed ("nccl" == RCCL) with torch.cuda.ExternalStream;
* `bench.py --gpus 2` launching two ranks by itself (both on device 0, collectives through gloo on host copies: RCCL
  refuses two ranks on one device) and reporting n_gpus = 2.
What mpi_dot_product / libsharp2's exchange do in the reference (comm_utils.f90:599-614); SURVEY.md 8e."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["native", "native-after-torch", "split", "torch-stream", "sliced"])
def test_rccl_one_rank_bit_equal(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), mode], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "RCCL_WORKER_OK " + mode in p.stdout, (p.returncode, p.stdout[-500:], p.stderr[-3000:])


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


def test_bench_force_dist_native_rccl_one_rank():
    """The N > 1 code path of bench.py itself (torch 'nccl' group for barrier / id broadcast, the library's own RCCL
    for the all-reduce) with a world of one rank."""
    env = dict(os.environ, CMDR_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29578")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                        "--config", "cfg2", "--nside", "64", "--lmax", "128", "--no-extras"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and lines[0]["rccl_world_size"] == 1 and lines[0]["collective"] == "rccl-native"
