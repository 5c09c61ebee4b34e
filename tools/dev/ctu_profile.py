"""dev: phase breakdown of the CTU search kernel (build csrc with EXTRA=-DCTU_PROFILE first)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch, time
from uvg266_amd import lib, api, layout
hip = lib.init(0)
W, H, depth, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
P = api.ctu_params(W, H, 22)
src = [tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in layout.synthetic_yuv420(W, H, t, depth)) for t in range(n)]
cs = api.CtuSearch(P, src)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time(); cs.run(); torch.cuda.synchronize(); dt = time.time() - t
    print(f"{n} pictures {W}x{H} {depth}-bit: {dt*1e3:.1f} ms -> {n/dt:.2f} pictures/s")
ctus = ((W + 63) // 64) * ((H + 63) // 64) * n
al = lambda v, a: (v + a - 1) // a * a
off_order = al(512 + 32768 + ctus * 4, 256); off_pics = al(off_order + ctus * 4, 256); off_scr = al(off_pics + n * 96, 256)
n_slots = min(2048, al(ctus, 32))          # a slot keeps the profile of the last CTU that ran on it
ws = cs.ws.cpu().numpy()
SZ = 68160
prof = np.stack([ws[off_scr + i * SZ + SZ - 1024: off_scr + i * SZ + SZ].view(np.uint64).reshape(4, 32) for i in range(n_slots)]).astype(np.float64)
lf = np.stack([ws[off_scr + i * SZ + SZ - 1024 - 128: off_scr + i * SZ + SZ - 1024].view(np.uint64) for i in range(n_slots)]).astype(np.float64)
lf = lf[prof[:, 0, 11] > 0]
prof = prof[prof[:, 0, 11] > 0]
names = ["rough search", "refs+predict", "residual+transforms+recon", "RDOQ", "SSD", "RD cost bits", "unpark/models", "64x64 candidate", "coder pass", "load", "store", "TOTAL",
         "rq: candidates+last", "rq: pre-walk", "rq: decide", "rq: accumulate+group", "rq: copy-out", "rq: cbf+last search", "rq: signs", "leaf depth 0", "leaf depth 1", "leaf depth 2", "leaf depth 3", "leaf depth 4 (4x4)", "rs: setup", "rs: rough_costs", "rs: mode cost", "rs: select", "cb: last+flags", "cb: records+budget", "cb: sweeps", "cb: bypass+lane0"]
tot = prof[:, 0, 11].mean()
print("mean cycles per CTU: total %.0f (%.2f ms at 2.1 GHz)" % (tot, tot / 2.1e6))
print("%-28s %12s %12s %12s %12s" % ("", "wave0 (4x4+walk)", "wave1 (8x8)", "wave2 (16x16)", "wave3 (32x32)"))
for i, nm in enumerate(names):
    print("  %-26s" % nm + "".join("%12.0f" % prof[:, w, i].mean() for w in range(4)))

print("chroma helper per CTU: Cb blocks handed to depth 3's wave %.1f, kept by the walk (wave busy) %.1f, the walk's wait for them %.0f cycles" % (prof[:, 1, 19].mean(), prof[:, 1, 20].mean(), prof[:, 1, 21].mean()))

lfn = ["area source load", "refs (luma)", "src->sgpr, mpm, planar/DC", "pass A (modes 4..65)", "selection (+ pass B)", "recon: refs (chroma)", "recon: predict + residual + DCT", "recon: RDOQ", "recon: dequant + IDCT + store + SSD", "fill_cu + help + between", "-", "bits: flags + mode bits (lane 0)", "bits: tr_cost (cbf + coeff_bits4)", "bits: cost + deblock marks"]
print("4x4 leaf, cycles per CTU (walk's wave):")
for i, nm in enumerate(lfn):
    if nm != "-": print("  %-44s %10.0f   per CU %7.0f" % (nm, lf[:, i].mean(), lf[:, i].mean() / 256))
