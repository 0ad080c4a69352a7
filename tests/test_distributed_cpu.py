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
