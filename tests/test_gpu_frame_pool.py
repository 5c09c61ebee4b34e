"""uvghip_frame_pool_* over the C ABI (api.FramePool): pictures from host memory one by one, in groups -- every picture and every WPP row
equals what the loop plan gives for the same picture on resident buffers (api.ClosedLoop, itself held to the reference encoder's records
in tests/test_gpu_closed_loop.py / test_gpu_slice_coder.py), whatever the grouping: a full group, a group cut short by a finish, by
another QP, slots reused out of order.  tests/test_gpu_dropin_frame.py runs the same entry points inside the reference encoder."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def loop_plan_result(params, yuv):
    """One picture through api.ClosedLoop -> ((y, u, v) numpy, [row bytes])."""
    import torch
    from uvg266_amd import api
    cl = api.ClosedLoop(params, [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv)])
    cl.run()
    torch.cuda.synchronize()
    rows, nb = cl.slice_data()
    rows, nb = rows.cpu().numpy(), nb.cpu().numpy()
    return tuple(t.cpu().numpy() for t in cl.out[0]), [rows[0, r, :nb[0, r]].tobytes() for r in range(nb.shape[1])]


@pytest.mark.parametrize("depth,w,h", [(8, 264, 136), (10, 200, 136)])
def test_pictures_through_the_pool_equal_the_loop_plans(hip, depth, w, h):
    from uvg266_amd import api
    pics = [H.varied_picture(w, h, t, depth) for t in range(7)]
    qps = [27, 27, 27, 32, 32, 27, 22]                          # pictures 3 and 5 and 6 arrive with other parameters: their groups are cut
    P = [api.ctu_params(w, h, q) for q in qps]
    want = [loop_plan_result(P[i], pics[i]) for i in range(7)]
    pool = api.FramePool(P[0], depth, n_slots=5, group_max=3)
    slot_of = [0, 1, 2, 3, 4, 0, 2]                             # slots 0 and 2 are reused once their pictures are finished

    def check(i, got):
        (y, u, v), rows = got
        for a, b in zip((y, u, v), want[i][0]):
            assert np.array_equal(a, b), f"picture {i}"
        assert rows == want[i][1], f"picture {i}: rows"

    for i in range(5):
        pool.begin(slot_of[i], P[i], pics[i])                   # 0 1 2 fill a group (launched by the third begin); 3 4 collect
    check(0, pool.finish(0))
    check(2, pool.finish(2))                                    # out of order inside a launched group
    pool.begin(slot_of[5], P[5], pics[5])                       # QP 27 behind the open QP 32 group: that group is launched, 5 opens the next
    pool.begin(slot_of[6], P[6], pics[6])                       # QP 22: 5's group is launched, 6 alone in the open one
    check(6, pool.finish(slot_of[6]))                           # asked for while its group still collects: launched by finish
    check(1, pool.finish(1))
    check(3, pool.finish(3))
    check(4, pool.finish(4))
    check(5, pool.finish(slot_of[5]))


def test_the_pool_refuses_misuse(hip):
    from uvg266_amd import api, lib
    w, h = 136, 72
    P = api.ctu_params(w, h, 27)
    pic = H.varied_picture(w, h, 0, 8)
    pool = api.FramePool(P, 8, n_slots=2, group_max=2)
    with pytest.raises(Exception, match="no picture has been begun"):
        pool.finish(1)
    pool.begin(0, P, pic)
    with pytest.raises(Exception, match="has not been finished"):
        pool.begin(0, P, pic)
    with pytest.raises(Exception):
        pool.begin(1, api.ctu_params(w + 8, h, 27), pic)        # the size is the pool's for good
    pool.finish(0)
    with pytest.raises(Exception):
        api.FramePool(P, 8, n_slots=1, group_max=1, sao_type=0)  # what uvghip_loop_plan_create refuses is refused at creation


def test_the_pools_pictures_and_rows_complete_the_encoders_stream(hip):
    """Anchored on the reference encoder's own file (tests/golden/ref_stream_192x128_8_qp27_3frames.npz): the three pictures through a pool
    of two slots; the hash of each returned picture (uvghip_picture_checksum) and its rows through uvghip_write_picture_nals, behind the
    encoder's parameter sets = the encoder's .266."""
    import ctypes
    import torch
    from uvg266_amd import api, layout, lib
    g = H.ctu_golden("ref_stream_192x128_8_qp27_3frames")
    w, h, depth, qp = (int(a) for a in g["meta"])
    P = api.ctu_params(w, h, qp)
    pool = api.FramePool(P, depth, n_slots=2, group_max=2)
    L = lib.init(0)
    stream = g["bitstream"].tobytes()
    out = stream[:stream.find(b"\x00\x00\x01\x00\x41")]
    pics = [layout.synthetic_yuv420(w, h, int(t), depth) for t in g["ts"]]
    pool.begin(0, P, pics[0])
    pool.begin(1, P, pics[1])
    done = [pool.finish(0)]
    pool.begin(0, P, pics[2])
    done += [pool.finish(1), pool.finish(0)]
    for poc, ((y, u, v), rows) in enumerate(done):
        sums = api.picture_checksum(*(torch.from_numpy(p).cuda() for p in (y, u, v))).cpu().numpy().view(np.uint32)
        sizes = np.array([len(r) for r in rows], np.int32)
        rows_2d = np.zeros((len(rows), int(sizes.max())), np.uint8)
        for r, b in enumerate(rows):
            rows_2d[r, :len(b)] = np.frombuffer(b, np.uint8)
        cap = int(sizes.sum()) + 64 + 4 * len(sizes)
        buf = np.zeros(cap, np.uint8)
        n = ctypes.c_size_t(0)
        ck = np.ascontiguousarray(sums, np.uint32)
        lib.check(L.uvghip_write_picture_nals(poc, 1, H.ptr(rows_2d), rows_2d.shape[1], H.ptr(sizes), len(sizes), H.ptr(ck), H.ptr(buf), cap, ctypes.byref(n)),
                  "uvghip_write_picture_nals")
        out += buf[:n.value].tobytes()
    assert out == stream


@pytest.mark.parametrize("depth,w,h,grid", [(8, 264, 136, ([2, 3], [1, 2])), (10, 328, 200, ([3, 3], [2, 2]))])
def test_tiled_pictures_through_the_pool_equal_the_tiles_plans(hip, depth, w, h, grid):
    """uvghip_frame_pool_create_tiles: a group is one tiles plan -- the pictures and ALL tiles' substreams in the order of the bitstream equal
    what api.TiledLoop gives for the same pictures on resident buffers (itself held to the reference's --tiles files, tests/test_gpu_tiles.py);
    three pictures through two slots, the third alone in its group."""
    import torch
    from uvg266_amd import api
    P = api.ctu_params(w, h, 27)
    pics = [H.varied_picture(w, h, t, depth) for t in (2, 1003, 5)]

    want = []
    for yuv in pics:
        tl = api.TiledLoop(P, [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in yuv)], grid)
        tl.run()
        lens, data, _ = tl.substreams()
        at = np.concatenate([[0], np.cumsum(lens[0])])
        torch.cuda.synchronize()
        want.append((tuple(t.cpu().numpy() for t in tl.out[0]),
                     [data[at[r]:at[r + 1]].tobytes() for r in range(lens.shape[1])]))
    pool = api.FramePool(P, depth, n_slots=2, group_max=2, tiles=grid)
    pool.begin(0, P, pics[0])
    pool.begin(1, P, pics[1])
    got = [pool.finish(0)]
    pool.begin(0, P, pics[2])
    got += [pool.finish(1), pool.finish(0)]
    for i, ((y, u, v), rows) in enumerate(got):
        for a, b in zip((y, u, v), want[i][0]):
            assert np.array_equal(a, b), f"picture {i}"
        assert rows == want[i][1], f"picture {i}: substreams"
