"""Frame sharding used by bench.py --gpus N, exercised with 2 gloo processes on CPU."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uvg266_amd import layout  # noqa: E402


def test_frame_assignment_is_a_partition():
    for world in (1, 2, 4, 8):
        seen = []
        for rank in range(world):
            seen += layout.frames_of_rank(rank, world, 60)
        assert sorted(seen) == list(range(60))
        sizes = [len(layout.frames_of_rank(r, world, 60)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = layout.frames_of_rank(rank, world, 37)
    # every rank "processes" its frames: here a checksum of the synthetic frame stands in for the kernels
    local = torch.zeros(37, dtype=torch.int64)
    for t in mine:
        y, u, v = layout.synthetic_yuv420(64, 48, t)
        local[t] = int(y.sum()) + int(u.sum()) + int(v.sum())
    dist.barrier()
    elapsed = torch.tensor([0.25 + rank], dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)          # bench.py takes the max over ranks
    dist.all_reduce(local, op=dist.ReduceOp.SUM)            # test-only gather: no rank overlaps, none missing
    if rank == 0:
        q.put((float(elapsed), local.tolist()))
    dist.destroy_process_group()


def test_two_process_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    elapsed, sums = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert elapsed == 1.25
    want = []
    for t in range(37):
        y, u, v = layout.synthetic_yuv420(64, 48, t)
        want.append(int(y.sum()) + int(u.sum()) + int(v.sum()))
    assert sums == want
