"""dev: the random-access clip of bench.py (ra_clip) at several numbers of pictures in flight; per-picture solo times.
usage: ra_clip_probe.py [frames] [in_flight ...] (0: by level)"""
import sys, os, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as Hh
from uvg266_amd import lib, api
lib.init(0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 33
flights = [int(a) for a in sys.argv[2:]] or [1, 32]
g = np.load(os.path.join(ROOT, "tests", "golden", "ref_gop16_states_qp27_65frames.npz"))
W, H, depth = 1920, 1080, 8
states = Hh.frame_states_from_records(g["meta"], g["lam"], g["refs"])[:frames]
display = [int(a) for a in g["display"][:frames]]
shown = {t: tuple(torch.from_numpy(np.ascontiguousarray(p)).cuda() for p in Hh.clip_picture(W, H, t, depth)) for t in sorted(set(display))}
loop = api.LowDelayLoop(W, H, depth, 1, states, [[shown[display[f]] for f in range(frames)]])
loop.run(); torch.cuda.synchronize()
import types
_orig = loop.L.uvghip_loop_pb_run
_times = []
class _Wrap:
    def __init__(self, L): self.__dict__["L"] = L
    def __getattr__(self, n):
        f = getattr(self.__dict__["L"], n)
        if n != "uvghip_loop_pb_run": return f
        def timed(*a):
            t = time.perf_counter(); r = f(*a); _times.append(time.perf_counter() - t); return r
        return timed
loop.L = _Wrap(loop.L)
for k in flights:
    _times.clear()
    if k == 0:
        lv = api.LowDelayLoop(W, H, depth, 1, states, [[shown[display[f]] for f in range(frames)]], by_level=True)
        lv.run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); lv.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{frames} pictures, by level ({len(lv.order)} launches, {1 + max(lv.level)} levels): enqueue {t1 - t0:.2f} s, total {t2 - t0:.2f} s -> {frames / (t2 - t0):.2f} pictures/s", flush=True)
        del lv
        continue
    t0 = time.perf_counter(); loop.run(in_flight=k); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host seconds per uvghip_loop_pb_run:", " ".join(f"{t:.2f}" for t in _times), flush=True)
    print(f"{frames} pictures, {k} in flight: enqueue {t1 - t0:.2f} s, total {t2 - t0:.2f} s -> {frames / (t2 - t0):.2f} pictures/s", flush=True)
# solo time of every picture (its references are there from the runs above)
st = torch.cuda.current_stream().cuda_stream
import ctypes
for f, step in enumerate(loop.steps[:0]):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if step[0] == "I": step[1].run(st)
    else: lib.check(loop.L.uvghip_loop_pb_run(depth, ctypes.byref(step[1]), 1, loop.sao_type, step[2].data_ptr(), st), "run")
    torch.cuda.synchronize()
    print(f"coded {f} poc {states[f]['poc']} type {states[f]['slice_type']} qp {states[f]['qp']} refs {states[f]['n_refs']}: {1e3 * (time.perf_counter() - t0):.0f} ms", flush=True)
