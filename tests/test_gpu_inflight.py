"""Pictures IN FLIGHT behind their references (uvghip_loop_pb_run_inflight; api.LowDelayLoop(inflight=True)): the encoder's --owf schedule
(src/encoderstate.c:1060-1116) -- CTU (x, y) of a picture starts when CTU (x + 2, y + 1) of the pictures it reads is final, the in-loop
filters (deblocking, SAO statistics / decision / reconstruction) run per CTU inside the persistent search kernel, ALL P / B pictures of a
clip share one launch.  The vectors are restricted to what is final in a reference that is still being coded (inflight_margin 11 =
fracmv_within_tile with cfg.owf != 0, search_inter.c:94-149): the stream is the one the reference writes with --owf != 0.
  * ref_inter_136x200_8_qp27_11frames_owf1 is such a run on content where the restriction bites (every picture, every row's bytes);
  * on the other goldens' content an --owf 1 run of the reference wrote the --owf 0 stream byte for byte (the sweep
    tools/refcheck/sweep_gop.py draws --owf 0 / 1): the restriction never decides, so the in-flight schedule must reproduce them too --
    low delay, random access (--gop 16 / 8), an open GOP with CRA pictures in the middle, 8 and 10 bit, 1080p by CRC."""
import os
import zlib
import numpy as np
import pytest
import helpers as H

pytestmark = pytest.mark.gpu


def _frame_rows(g):
    frames = int(g["dims"][4])
    first = {}
    for k in range(len(g["meta"])):
        first.setdefault(int(g["meta"][k][0]), k)
    ks = [first[f] for f in range(frames)]
    return g["meta"][ks], g["lam"][ks], g["refs"][ks]


CASES = [("ref_inter_136x200_8_qp27_11frames_owf1", 1, 3), ("ref_inter_136x200_8_qp27_11frames_owf1", 3, 3), ("ref_inter_264x136_8_qp32_9frames", 2, 3),
         ("ref_inter_136x72_10_qp22_4frames", 1, 3), ("ref_inter_192x128_8_qp17_5frames", 1, 3), ("ref_inter_136x72_8_qp27_17frames_ra16", 2, 3),
         ("ref_inter_136x72_10_qp22_17frames_ra16", 1, 3), ("ref_inter_136x72_8_qp27_9frames_ra8", 1, 3), ("ref_inter_136x72_8_qp27_33frames_ra16p16", 1, 3),
         ("ref_intercrc_136x72_8_qp27_65frames_ra16", 1, 3), ("ref_intercrc_1920x1080_8_qp27_5frames", 1, 3), ("ref_intercrc_1920x1080_8_qp27_17frames_ra16", 1, 3), ("ref_intercrc_1920x1080_10_qp32_3frames", 1, 3),
         # BASELINE configs[3]'s geometry and depth: 3840x2160 10-bit, low delay and --gop 16 (the four-wave build's 10-bit LDS image)
         ("ref_intercrc_3840x2160_10_qp27_3frames", 1, 3), ("ref_intercrc_3840x2160_10_qp27_17frames_ra16", 1, 3)]


# the I pictures in the flight too: searched by a few persistent workgroups on a second stream, filtered CTU by CTU by the in-flight launch
INTRA_CASES = [("ref_inter_136x200_8_qp27_11frames_owf1", 1), ("ref_inter_264x136_8_qp32_9frames", 2), ("ref_inter_136x72_10_qp22_17frames_ra16", 1),
               ("ref_inter_136x72_8_qp27_33frames_ra16p16", 1), ("ref_intercrc_1920x1080_8_qp27_5frames", 1), ("ref_intercrc_1920x1080_8_qp27_17frames_ra16", 1),
               ("ref_intercrc_1920x1080_10_qp32_3frames", 1)]


@pytest.mark.parametrize("name,n_seq", INTRA_CASES)
def test_intra_pictures_in_the_flight(hip, name, n_seq):
    test_pictures_in_flight_reproduce_the_encoder(hip, name, n_seq, 3, intra_in_flight=True)


@pytest.mark.parametrize("name,n_seq,sao_type", CASES)
def test_pictures_in_flight_reproduce_the_encoder(hip, name, n_seq, sao_type, intra_in_flight=False):
    import time
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    hc = (Hh + 63) // 64
    crc_only = "final_crc" in g.files
    meta, lam, refs = (g["meta"], g["lam"], g["refs"]) if crc_only else _frame_rows(g)
    states = H.frame_states_from_records(meta, lam, refs)
    cfg = [int(a) for a in g["cfg"]] if "cfg" in g.files else [1, 6, 2, 1, 4, 1, 0, 0]
    pics = H.golden_sources(g)
    src = [[tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pics[f]) for f in range(frames)] for _ in range(n_seq)]
    loop = api.LowDelayLoop(W, Hh, depth, n_seq, states, src, sao_type=sao_type, tmvp=cfg[0], max_merge=cfg[1], merge_level=cfg[2], bipred=cfg[3], fme_level=cfg[4],
                            early_skip=cfg[5], rd=cfg[6] if len(cfg) > 6 else 0, inflight=True, inflight_margin=11 if sao_type else 9, intra_in_flight=intra_in_flight)
    for rep in range(2):          # twice: the second run starts from a used workspace
        t0 = time.time()
        loop.run()
        torch.cuda.synchronize()
        print(f"{name}: {frames} pictures x {n_seq} in flight: {time.time() - t0:.2f} s")
        for f in range(frames):
            rows, nb = loop.rows[f].cpu().numpy(), loop.row_bytes[f].cpu().numpy()
            for s in range(n_seq):
                planes = [a.cpu().numpy() for a in loop.out[f][s]]
                if crc_only:
                    assert zlib.crc32(b"".join(np.ascontiguousarray(a).tobytes() for a in planes)) == int(g["final_crc"][f]), (name, "picture", f, "sequence", s)
                    for r in range(hc):
                        assert nb[s, r] == int(g["row_len"][f * hc + r]) and zlib.crc32(rows[s, r, :nb[s, r]].tobytes()) == int(g["row_crc"][f * hc + r]), (name, f, s, "row", r)
                else:
                    for cidx, nme in enumerate(("final_y", "final_u", "final_v")):
                        assert np.array_equal(planes[cidx], g[nme][f]), (name, rep, f, s, nme, np.argwhere(planes[cidx] != g[nme][f])[:4].tolist())
                    off = g["row_off"][f * hc:f * hc + hc + 1]
                    for r in range(hc):
                        want = g["row_bytes"][off[r]:off[r + 1]]
                        assert nb[s, r] == len(want) and np.array_equal(rows[s, r, :nb[s, r]], want), (name, f, s, "row", r)


def test_inflight_refuses_what_it_cannot_keep(hip):
    """A reference inside the call needs the vector restriction (inflight_margin 11 / 9) and an earlier index."""
    import ctypes
    import torch
    from uvg266_amd import api
    g = np.load(os.path.join(H.GOLDEN, "ref_inter_136x72_10_qp22_4frames.npz"))
    W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
    meta, lam, refs = _frame_rows(g)
    states = H.frame_states_from_records(meta, lam, refs)
    pics = H.golden_sources(g)
    src = [[tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pics[f]) for f in range(frames)]]
    loop = api.LowDelayLoop(W, Hh, depth, 1, states, src, inflight=True, inflight_margin=0)
    with pytest.raises(RuntimeError, match="inflight_margin"):
        loop.run()
    torch.cuda.synchronize()
