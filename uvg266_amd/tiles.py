"""The tiles of ONE all-intra picture over the ranks of a node (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

Under the closed loop the CTU rows of a picture are a pipeline (a band waits for the band above: DESIGN 6), the tiles of --tiles are not:
the reference gives a tile no neighbour in the search and no sample in the in-loop filters (pps_loop_filter_across_tiles_enabled_flag = 0,
src/encoder_state-bitstream.c:788), so every rank runs its tiles from the first CTU on and the only exchange is at the end of a group of
pictures -- per picture the substreams' lengths, their bytes and three words of checksum, to the rank that writes the NAL units:

    owner  = assign(rects, world)                          # the same on every rank
    loop   = api.TiledLoop(params, src, grid, owned=owner == rank);  loop.run()
    nals   = gather_nals(loop.substreams(), first_poc, sao=True)      # list of bytes on rank `dst`, None elsewhere

gather_nals moves: one all-gather of [count, n_substreams] int32 lengths + [count, 3] checksum terms, one all-gather of the ranks' bytes padded
to the longest contribution (a 1080p picture at QP 22: ~0.3 MB in all).  Nothing else crosses xGMI: no halo, no reference picture (all-intra).
"""
import ctypes

import numpy as np

from . import lib as _lib


def assign(rects, world):
    """owner[tile] in [0, world): the tiles of a grid (api.tile_grid's rects) over `world` ranks, balanced by what bounds a tile's time under
    WPP -- its diagonals, columns + 2 (rows - 1) CTUs -- with the CTU count as the tie-breaker: longest first onto the least loaded rank.
    Deterministic: every rank computes the same table."""
    rects = np.asarray(rects)
    wc, hc = (rects[:, 2] + 63) // 64, (rects[:, 3] + 63) // 64
    cost = (wc + 2 * (hc - 1)).astype(np.int64) * 4096 + wc * hc
    owner = np.zeros(len(rects), np.int32)
    load = np.zeros(world, np.int64)
    for t in sorted(range(len(rects)), key=lambda t: (-int(cost[t]), t)):
        r = int(np.argmin(load))          # (the first of equals)
        owner[t] = r
        load[r] += cost[t]
    return owner


def write_nals(lens, pieces, sums, first_poc=0, sao=True):
    """The NAL units of `count` pictures from everything the ranks contributed: lens [world, count, n_substreams] (each substream non-zero on
    exactly one rank), pieces[rank] = that rank's bytes (owned substreams in bitstream order, pictures one after the other), sums
    [world, count, 3].  -> list of bytes (uvghip_write_picture_nals, a host function of the library)."""
    L = _lib.load_library()
    lens = np.asarray(lens, np.int64)
    world, count, n_sub = lens.shape
    if not ((lens > 0).sum(axis=0) == 1).all():
        raise ValueError("every substream needs exactly one owner")
    total = lens.sum(axis=0).astype(np.int32)                         # [count, n_sub]
    ck = (np.asarray(sums, np.uint64).sum(axis=0) & 0xFFFFFFFF).astype(np.uint32)
    at = [0] * world
    out = []
    for i in range(count):
        pitch = int(total[i].max())
        rows = np.zeros((n_sub, pitch), np.uint8)
        for s in range(n_sub):
            r = int(np.argmax(lens[:, i, s]))
            n = int(lens[r, i, s])
            rows[s, :n] = pieces[r][at[r]:at[r] + n]
            at[r] += n
        cap = int(total[i].sum()) + 64 + 4 * n_sub
        buf = np.zeros(cap, np.uint8)
        n = ctypes.c_size_t(0)
        sizes = np.ascontiguousarray(total[i])
        c3 = np.ascontiguousarray(ck[i])
        _lib.check(L.uvghip_write_picture_nals(first_poc + i, 1 if sao else 0, rows.ctypes.data, pitch, sizes.ctypes.data, n_sub, c3.ctypes.data, buf.ctypes.data, cap,
                                               ctypes.byref(n)), "uvghip_write_picture_nals")
        out.append(buf[:n.value].tobytes())
    for r in range(world):
        if at[r] != len(pieces[r]):
            raise ValueError("a rank's bytes do not match its lengths")
    return out


def gather_nals(contribution, first_poc=0, sao=True, group=None, dst=0):
    """contribution = (lens [count, n_substreams] int32, bytes uint8, sums [count, 3] uint32) of THIS rank (api.TiledLoop.substreams()).
    Collective over `group`; -> the pictures' NAL units (list of bytes) on rank `dst`, None on the others."""
    import torch
    import torch.distributed as dist
    lens, data, sums = contribution
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    count, n_sub = lens.shape
    # lengths and checksum terms in one tensor, the bytes' length behind them
    head = torch.from_numpy(np.concatenate([np.asarray(lens, np.int64).reshape(-1), np.asarray(sums, np.int64).reshape(-1), [len(data)]])).to(dev)
    heads = [torch.empty_like(head) for _ in range(world)]
    dist.all_gather(heads, head, group=group)
    heads = [h.cpu().numpy() for h in heads]
    longest = max(int(h[-1]) for h in heads)
    mine = torch.zeros(max(longest, 1), dtype=torch.uint8)
    mine[:len(data)] = torch.from_numpy(np.ascontiguousarray(data))
    mine = mine.to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    if rank != dst:
        return None
    all_lens = np.stack([h[:count * n_sub].reshape(count, n_sub) for h in heads])
    all_sums = np.stack([h[count * n_sub:count * n_sub + 3 * count].reshape(count, 3) for h in heads])
    pieces = [p.cpu().numpy()[:int(h[-1])] for p, h in zip(parts, heads)]
    return write_nals(all_lens, pieces, all_sums, first_poc, sao)
