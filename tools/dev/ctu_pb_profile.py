"""dev: phase breakdown of the P / B CTU search kernel (build csrc/ctu_search_pb with EXTRA=-DCTU_PROFILE first): the 1080p low-delay
golden's sequence, one sequence, cycles per CTU of the last B picture group."""
import sys, os, ctypes
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch, time
import helpers as H
from uvg266_amd import lib, api
L = lib.init(0)
name = sys.argv[1] if len(sys.argv) > 1 else "ref_intercrc_1920x1080_8_qp27_5frames"
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = np.load(os.path.join(H.GOLDEN, name + ".npz"))
W, Hh, depth, qp0, frames = (int(a) for a in g["dims"])
if "final_crc" in g.files:
    meta, lam, refs = g["meta"], g["lam"], g["refs"]
else:
    first = {}
    for k in range(len(g["meta"])):
        first.setdefault(int(g["meta"][k][0]), k)
    ks = [first[f] for f in range(frames)]
    meta, lam, refs = g["meta"][ks], g["lam"][ks], g["refs"][ks]
states = H.frame_states_from_records(meta, lam, refs)
frames = int(os.environ.get("PROFILE_FRAMES", frames))
states = states[:frames]
pics = H.golden_sources(g)[:frames] if int(os.environ.get("PROFILE_GOLDEN_SOURCES", "1")) else [H.moving_picture(W, Hh, t, depth) for t in range(frames)]
one = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in pics[f]) for f in range(frames)]
loop = api.LowDelayLoop(W, Hh, depth, n_seq, states, [one] * n_seq)
torch.cuda.synchronize(); t = time.time(); loop.run(); torch.cuda.synchronize()
print(f"{n_seq} x {frames} pictures {W}x{Hh}: {time.time() - t:.2f} s")
L.uvghip_ctu_search_pb_debug_scratch.restype = ctypes.c_size_t
sb, ns = ctypes.c_size_t(), ctypes.c_int()
names = ["candidate lists", "merge analysis", "early skip test", "integer ME", "fractional ME", "bi-prediction", "intra rough + chroma trial", "inter CU: pred + residual",
         "inter CU: bits + cost", "intra CU (eval_cu)", "unpark / save64 / restore64", "load", "store + deblock side effect", "coder pass", "TOTAL", "4x4 leaves (inside the leaf wave's time)",
         "walk waits for the leaf wave", "walk waits for the depth wave", "aborts (count)", "depth wave: 16x16 evals", "depth wave: 32x32 evals", "leaf wave busy", "walk: CUs evaluated in place (64x64; every depth in the one-wave build)", "walk: 8x8 CUs beside the leaf wave"]
for f, step in enumerate(loop.steps):
    if step[0] != "PB":
        continue
    ws = step[2].cpu().numpy()
    off = L.uvghip_ctu_search_pb_debug_scratch(n_seq, W, Hh, ctypes.byref(sb), ctypes.byref(ns))
    SZ = sb.value
    prof = np.stack([ws[off + i * SZ + SZ - 1024 - 128 - 192: off + i * SZ + SZ - 1024 - 128].view(np.uint64) for i in range(ns.value)]).astype(np.float64)
    prof = prof[prof[:, 14] > 0]
    tot = prof[:, 14].mean()
    print(f"picture {f}: {len(prof)} slots, mean cycles per CTU {tot:.0f} ({tot / 1e5:.2f} ms at 100 MHz s_memtime)")
    for i, nm in enumerate(names):
        print("   %-30s %12.0f  %5.1f %%" % (nm, prof[:, i].mean(), 100 * prof[:, i].mean() / tot))
    old = np.stack([ws[off + i * SZ + SZ - 1024: off + i * SZ + SZ].view(np.uint64).reshape(4, 32)[0] for i in range(ns.value)]).astype(np.float64)
    old = old[old[:, 1].astype(bool) | old[:, 3].astype(bool)]
    onames = ["rough search", "refs+predict", "residual+transforms+recon", "RDOQ", "SSD", "RD cost bits (eval_cu)", "unpark/models", "64x64 cand", "coder", "load", "store", "total(intra)",
              "rq: candidates+last", "rq: pre-walk", "rq: decide", "rq: accumulate+group", "rq: copy-out", "rq: cbf+last search", "rq: signs", "-", "-", "-", "-", "-", "rs: setup", "rs: rough_costs",
              "rs: mode cost", "rs: select", "cb: last+flags", "cb: records+budget", "cb: sweeps", "cb: bypass+lane0"]
    print("   -- ctu_core.h counters of the same CTUs (all callers) --")
    for i, nm in enumerate(onames):
        if old[:, i].mean() > 0:
            print("   %-30s %12.0f  %5.1f %%" % (nm, old[:, i].mean(), 100 * old[:, i].mean() / tot))
