"""CPU tier: the launch logic of bench.py -- `--gpus N` starts N fresh rank processes itself, never silently runs one
GPU -- rehearsed end to end through the host emulation of the library and gloo (CMDR_BENCH_REHEARSE_EMUL=1; the JSON
line says "REHEARSAL", it is not a measurement).  The same code path drives RCCL on the GPU box
(tests/test_rccl_gpu.py)."""
import json
import os
import subprocess
import sys

from helpers import ROOT, emul_lib

ARGS = ["--steps", "1", "--warmup", "0", "--config", "cfg2", "--nside", "16", "--lmax", "32", "--no-extras"]


def run_bench(extra, env_extra=None, expect_rc=0):
    env = dict(os.environ, CMDR_BENCH_REHEARSE_EMUL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + ARGS, env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == expect_rc, (p.returncode, p.stderr[-2000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return [json.loads(ln) for ln in lines], p


def test_gpus_flag_launches_that_many_ranks():
    emul_lib()   # build once before the ranks race for it
    one, _ = run_bench(["--gpus", "1"])
    two, _ = run_bench(["--gpus", "2"])
    four, _ = run_bench(["--gpus", "4"], {"CMDR_BENCH_SHARD": "2x2"})      # 2 band groups x 2 ring sets
    assert len(one) == len(two) == len(four) == 1                          # ONE JSON line, from rank 0
    assert one[0]["n_gpus"] == 1 and two[0]["n_gpus"] == 2 and four[0]["n_gpus"] == 4
    assert two[0]["config"]["ring_parts"] == 2 and two[0]["config"]["band_parts"] == 1
    assert four[0]["config"]["ring_parts"] == 2 and four[0]["config"]["band_parts"] == 2
    assert "REHEARSAL" in two[0]["data"]
    for r in (two[0], four[0]):     # the sharded solves are the same solve: same preconditioned residual after 40 iterations
        assert abs(r["solve"]["delta0"] - one[0]["solve"]["delta0"]) < 1e-9 * one[0]["solve"]["delta0"]
        assert abs(r["solve"]["res"] - one[0]["solve"]["res"]) < 1e-6 * one[0]["solve"]["res"]
        assert r["solve"]["niter"] == 40


def test_world_size_mismatch_is_an_error_not_a_one_gpu_run():
    out, p = run_bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0"}, expect_rc=2)
    assert out == [] and "refusing" in p.stderr


def test_a_rank_that_dies_ends_the_run_promptly():
    """The parent polls its children: when one rank exits non-zero the others (blocked in a collective) are terminated
    and the launcher itself exits non-zero -- within seconds, not at a watchdog's or the driver's limit."""
    import time
    emul_lib()
    t0 = time.time()
    out, p = run_bench(["--gpus", "2"], {"CMDR_BENCH_TEST_DIE_RANK": "1"}, expect_rc=1)
    assert time.time() - t0 < 30, time.time() - t0
    assert out == [] and "ranks failed" in p.stderr and "terminated" in p.stderr
