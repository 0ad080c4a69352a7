"""CPU, gloo, world size 2: the N>1 control path of bench.py (independent prompts per rank, barriers, MAX
reduction of the elapsed time, whole-job throughput). The data path has no collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dist.barrier()
        local = 1.0 + 0.5 * rank                      # rank 1 is the slow one
        elapsed = bench.job_elapsed(local, dist, torch.device("cpu"))
        dist.barrier()
        q.put((rank, elapsed, bench.rank_seed(rank), [bench.rank_view(rank, i, 16) for i in range(20)],
               bench.job_throughput(world, 10, elapsed)))
    finally:
        dist.destroy_process_group()


def test_two_rank_timing_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, e0, s0, v0, t0), (r1, e1, s1, v1, t1) = res
    assert (r0, r1) == (0, 1)
    assert e0 == e1 == 1.5                             # MAX over ranks, identical on every rank
    assert s0 != s1                                    # independent seeds
    assert v0 != v1 and set(v0) <= set(range(16))      # different camera sequences, all valid views
    assert t0 == t1 == pytest.approx(2 * 10 / 1.5)     # whole-job aggregate, not per GPU


def test_single_process_helpers():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.job_elapsed(0.25, None, None) == 0.25
    assert bench.job_throughput(8, 40, 2.0) == 160.0
    assert bench.rank_seed(0) == 0 and bench.rank_seed(3) == 3000


def _run_bench_dry(world, extra_env=None, args=("--steps", "10", "--warmup", "2")):
    """bench.py --dry-run-cpu as the driver launches the real thing: one process per rank, torch.distributed.run."""
    import json
    import subprocess
    env = dict(os.environ, **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run-cpu", *args]
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run-cpu", *args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout            # rank 0 only, ONE JSON line
    return json.loads(lines[0])


def test_bench_main_control_path_two_ranks():
    """The N > 1 path of bench.main() end to end under gloo: prior agreement (MIN all-reduce), priming, the barrier-bracketed
    timed region with its phase boundary, MAX reduction of the elapsed time, whole-job throughput, rank-0-only output,
    destroy_process_group — everything but the GPU work (DryJob)."""
    r = _run_bench_dry(2)
    assert r["n_gpus"] == 2 and r["steps"] == 10 and r["warmup"] == 2 and r["dry_run"] is True
    assert r["scaling"] == "weak" and r["higher_is_better"] is True and r["metric"] == "sds_iters_per_sec"
    assert set(r["phases"]) == {"latent", "rgb"} and r["phases"]["latent"]["steps"] == 2 and r["phases"]["rgb"]["steps"] == 8
    assert r["value"] == pytest.approx(2 * 10 / (r["ms_per_step"] * 10 / 1e3), rel=1e-6)       # aggregate over both ranks
    assert r["optimizer_steps_applied"] == 10 and r["config"]["guidance"] == "none"
    assert r["config"]["parallelism"] == "independent-prompts x2"


def test_bench_main_control_path_eight_ranks():
    """The launch the driver uses on an 8-GPU node, minus the GPUs: 8 ranks under gloo through bench.main() — rank 0 builds its
    job first, the others behind a barrier (the serialised solver search), one time-to-first-barrier per rank, IF preset plan."""
    r = _run_bench_dry(8, args=("--steps", "6", "--warmup", "1", "--prior", "if"))
    assert r["n_gpus"] == 8 and r["config"]["parallelism"] == "independent-prompts x8"
    assert list(r["phases"]) == ["rgb"] and r["phases"]["rgb"]["steps"] == 6          # --IF: no latent phase
    assert r["value"] == pytest.approx(8 * 6 / (r["ms_per_step"] * 6 / 1e3), rel=1e-6)
    t = r["seconds_to_first_barrier_per_rank"]
    assert len(t) == 8 and all(x is not None and x >= 0 for x in t)
    assert min(t[1:]) >= t[0] - 0.05                   # nobody passed the barrier before rank 0 had built its job


def test_bench_ranks_agree_on_the_prior():
    """One rank failing to build the big prior must move EVERY rank to the synthetic one (same configuration, same barriers)."""
    r = _run_bench_dry(2, {"SDFX_DRY_PRIOR_FAIL_RANK": "1"})
    assert r["config"]["guidance"] == "synthetic"


def test_bench_single_phase_and_single_rank_dry():
    r = _run_bench_dry(1, args=("--steps", "5", "--warmup", "1", "--phase", "rgb"))
    assert r["n_gpus"] == 1 and list(r["phases"]) == ["rgb"] and r["phases"]["rgb"]["steps"] == 5


def test_phase_plan():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.phase_plan("mix", 40) == [("latent", 8), ("rgb", 32)]
    assert bench.phase_plan("mix", 1) == [("latent", 1)]
    assert bench.phase_plan("latent", 7) == [("latent", 7)] and bench.phase_plan("rgb", 7) == [("rgb", 7)]
    assert bench.phase_plan("mix", 40, "if") == [("rgb", 40)] and bench.phase_plan("latent", 5, "if") == [("rgb", 5)]
