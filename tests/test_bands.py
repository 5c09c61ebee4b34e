"""CTU-row bands: the C plan (host only), the exchange lists, and the schedule run by 2 and 3 gloo processes on CPU.

The bit-exactness of the banded FILTERS against the whole-frame kernels is a GPU test (tests/test_gpu_bands.py); here the
data movement itself is pinned: every halo row a kernel of rank r reads lies either in r's own rows or in what the
exchange delivers, every transfer pairs up with its peer, and the gather leaves the whole picture on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from uvg266_amd import bands  # noqa: E402


@pytest.mark.parametrize("height", [64, 264, 480, 1080, 2160])
def test_plan_partitions_the_ctu_rows(height):
    rows = (height + 63) // 64
    for n in (1, 2, 3, 4, 8):
        if n > rows:
            with pytest.raises(ValueError):
                bands.BandLayout(height, n, 0)
            continue
        plans = [bands.BandLayout(height, n, r) for r in range(n)]
        assert plans[0].y0 == 0 and plans[-1].y1 == height
        sizes = [p.ctu_row1 - p.ctu_row0 for p in plans]
        assert sum(sizes) == rows and max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
        for a, b in zip(plans, plans[1:]):
            assert a.y1 == b.y0 and a.down == b.rank and b.up == a.rank and a.y1 % 64 == 0
        assert plans[0].up == -1 and plans[-1].down == -1


def _planes(height, width, rank, tagged=True):
    """Full-size planes of one rank: owned rows carry (row, rank-independent) content, the rest a poison value."""
    def mk(h, w, sub, dtype=torch.int32):
        t = torch.full((h, w), -1, dtype=dtype)
        return t
    return dict(y=mk(height, width, 1), u=mk(height // 2, width // 2, 2), v=mk(height // 2, width // 2, 2),
                scu=mk((height + 3) // 4, (width // 4) * 8, 4))


def _fill_owned(lay, planes, gen):
    """Owned rows <- the reference content gen[name] (what the rank's kernels would have produced)."""
    for name, sub in (("y", 1), ("u", 2), ("v", 2), ("scu", 4)):
        r0, r1 = lay.owned(sub)
        planes[name][r0:r1] = gen[name][r0:r1]


def _reference(height, width, seed):
    g = torch.Generator().manual_seed(seed)
    return dict(y=torch.randint(0, 1 << 20, (height, width), generator=g, dtype=torch.int32),
                u=torch.randint(0, 1 << 20, (height // 2, width // 2), generator=g, dtype=torch.int32),
                v=torch.randint(0, 1 << 20, (height // 2, width // 2), generator=g, dtype=torch.int32),
                scu=torch.randint(0, 1 << 20, ((height + 3) // 4, (width // 4) * 8), generator=g, dtype=torch.int32))


def _check_after_deblock_halo(lay, planes, ref):
    """What the horizontal-edge pass + SAO of this band read: own rows, 4 rows above (P side), 8 below (Q side),
    chroma half, one SCU row either side -- all must hold the true content; everything else must still be poison."""
    H = lay.height
    lo = lay.y0 - (bands.HALO_DBK_P if lay.up >= 0 else 0)
    hi = min(H, lay.y1 + (bands.HALO_DBK_Q if lay.down >= 0 else 0))
    for name, sub in (("y", 1), ("u", 2), ("v", 2)):
        a, b = lo // sub, (hi + sub - 1) // sub
        assert torch.equal(planes[name][a:b], ref[name][a:b]), name
        assert (planes[name][:a] == -1).all() and (planes[name][b:] == -1).all(), name
    a = (lay.y0 - (4 if lay.up >= 0 else 0)) // 4
    b = (min(H, lay.y1 + (4 if lay.down >= 0 else 0)) + 3) // 4
    assert torch.equal(planes["scu"][a:b], ref["scu"][a:b])
    assert (planes["scu"][:a] == -1).all() and (planes["scu"][b:] == -1).all()


def _check_after_alf_halo(lay, planes, ref):
    lo = lay.y0 - (bands.HALO_ALF if lay.up >= 0 else 0)
    hi = min(lay.height, lay.y1 + (bands.HALO_ALF if lay.down >= 0 else 0))
    for name, sub in (("y", 1), ("u", 2), ("v", 2)):
        a, b = lo // sub, (hi + sub - 1) // sub
        assert torch.equal(planes[name][a:b], ref[name][a:b]), name
        assert (planes[name][:a] == -1).all() and (planes[name][b:] == -1).all(), name


@pytest.mark.parametrize("height,n", [(264, 2), (264, 4), (1080, 8), (2160, 8), (480, 3)])
def test_emulated_exchanges_deliver_exactly_the_halos(height, n):
    width = 64
    lays = [bands.BandLayout(height, n, r) for r in range(n)]
    ref = _reference(height, width, 1)
    # deblock halo
    P = [_planes(height, width, r) for r in range(n)]
    for lay, p in zip(lays, P):
        _fill_owned(lay, p, ref)
    bands.emulate([lay.halo_deblock(p["y"], p["u"], p["v"], p["scu"]) for lay, p in zip(lays, P)])
    for lay, p in zip(lays, P):
        _check_after_deblock_halo(lay, p, ref)
    # ALF halo
    P = [_planes(height, width, r) for r in range(n)]
    for lay, p in zip(lays, P):
        _fill_owned(lay, p, ref)
    bands.emulate([lay.halo_alf(p["y"], p["u"], p["v"]) for lay, p in zip(lays, P)])
    for lay, p in zip(lays, P):
        _check_after_alf_halo(lay, p, ref)
    # gather
    P = [_planes(height, width, r) for r in range(n)]
    for lay, p in zip(lays, P):
        _fill_owned(lay, p, ref)
    specs = [lay.gather(p["y"], p["u"], p["v"]) for lay, p in zip(lays, P)]
    bands.emulate(specs)
    for p in P:
        for name in "yuv":
            assert torch.equal(p[name], ref[name])
    # bytes: what a rank sends in the gather is its band to each of the n - 1 peers
    for lay, s in zip(lays, specs):
        sent, recvd = bands.spec_bytes(s)
        own = (lay.y1 - lay.y0) * width * 4 + 2 * ((lay.y1 + 1) // 2 - lay.y0 // 2) * (width // 2) * 4
        assert sent == own * (n - 1)
        assert recvd == (height * width * 4 + 2 * (height // 2) * (width // 2) * 4) - own


def _worker(rank, world, port, height, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        width = 64
        lay = bands.BandLayout(height, world, rank)
        tr = bands.TorchTransport(dist)
        ref = _reference(height, width, 5)          # same seed on every rank: the content a whole-frame run would hold
        ok = True
        p = _planes(height, width, rank)
        _fill_owned(lay, p, ref)
        tr.exchange(lay.halo_deblock(p["y"], p["u"], p["v"], p["scu"]))
        _check_after_deblock_halo(lay, p, ref)
        p = _planes(height, width, rank)
        _fill_owned(lay, p, ref)
        tr.exchange(lay.halo_alf(p["y"], p["u"], p["v"]))
        _check_after_alf_halo(lay, p, ref)
        p = _planes(height, width, rank)
        _fill_owned(lay, p, ref)
        tr.exchange(lay.gather(p["y"], p["u"], p["v"]))
        for name in "yuv":
            ok &= bool(torch.equal(p[name], ref[name]))
        # covariance sums: every rank contributes the sums of its own CTU rows
        part = torch.arange(25 * 1509, dtype=torch.int64).reshape(25, 1509) * (lay.ctu_row1 - lay.ctu_row0)
        tr.allreduce(part)
        ok &= bool(torch.equal(part, torch.arange(25 * 1509, dtype=torch.int64).reshape(25, 1509) * lay.ctu_rows))
        dist.barrier()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 264), (3, 1080)])
def test_gloo_processes_run_the_schedule(world, height):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31000 + (os.getpid() * 7 + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, height, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(r, True) for r in range(world)]


def _search_halo_worker(rank, nranks, port, height, q):
    import torch
    import torch.distributed as dist
    from uvg266_amd import bands
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=nranks)
    try:
        W, wc = 192, 3
        hc = (height + 63) // 64
        lay = bands.BandLayout(height, nranks, rank)
        mk = lambda shape, dt: torch.full(shape, -1, dtype=dt)
        y, u, v = mk((height, W), torch.int16), mk((height // 2, W // 2), torch.int16), mk((height // 2, W // 2), torch.int16)
        scu, mod = mk((hc * 16, wc * 16 * 32), torch.uint8), mk((wc * hc, 3 * 257), torch.int32)
        # what this rank "produced": its own rows carry rank + 1 in every table
        y[lay.y0:lay.y1] = rank + 1; u[lay.y0 // 2:lay.y1 // 2] = rank + 1; v[lay.y0 // 2:lay.y1 // 2] = rank + 1
        scu[lay.y0 // 4:lay.y1 // 4] = rank + 1; mod[lay.ctu_row0 * wc:lay.ctu_row1 * wc] = rank + 1
        tr = bands.TorchTransport(dist)
        tr.exchange(lay.halo_search(y, u, v, scu, mod, wc))
        ok = True
        if rank > 0:
            a = lay.y0
            ok &= bool((y[a - 1] == rank).all() and (u[a // 2 - 1] == rank).all() and (v[a // 2 - 1] == rank).all() and (scu[a // 4 - 1] == rank).all())
            ok &= bool((mod[(lay.ctu_row0 - 1) * wc] == rank).all())
            # and nothing else arrived: the line above the received one is still untouched
            ok &= bool(a < 2 or (y[a - 2] == -1).all())
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("nranks,height", [(2, 256), (3, 448)])
def test_search_halo_goes_down_one_band_over_gloo(nranks, height):
    """The halo of the row-sharded closed-loop search (bands.BandLayout.halo_search): every rank receives from the band above exactly the
    last reconstruction line, the last row of side information and the models of that band's last row's first CTU -- executed by real
    processes over gloo with the exchange lists the RCCL transport takes."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_search_halo_worker, args=(r, nranks, port, height, q)) for r in range(nranks)]
    for p in ps:
        p.start()
    got = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
    assert got == [(r, True) for r in range(nranks)]
