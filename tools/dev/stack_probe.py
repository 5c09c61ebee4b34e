"""Developer probe: per-picture plane kernels on one 1080p picture vs on F pictures stacked into one tall plane."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from uvg266_amd import api, lib, layout, pipeline
L = lib.init(0)
F = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H, depth = 1920, 1080, 8
P = lambda t: t.data_ptr()
st = torch.cuda.current_stream().cuda_stream
modes = api.make_modes(pipeline.MODES)
y1 = torch.randint(0, 256, (F * H, W), dtype=torch.uint8, device="cuda")
pred = torch.randint(0, 256, (F * H, W), dtype=torch.uint8, device="cuda")
rec = torch.zeros_like(y1)
def timeit(fn, args, reps=20):
    for _ in range(3): fn(*args, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn(*args, st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (32, 16, 8, 4):
    own = layout.intra_availability(layout.block_grid(W, H, n), n, W, H)
    cnt = len(own)
    allb = np.concatenate([own + np.array([0, f * H, 0, 0]) for f in range(F)])
    for tag, tab, k in (("1 pic", own, cnt), (f"{F} pics", allb, F * cnt)):
        blks, tus = api.make_intra_blocks(tab), api.make_tus(tab[:, :2])
        coef = torch.zeros((k, n, n), dtype=torch.int16, device="cuda")
        best = torch.zeros(k, dtype=torch.int8, device="cuda"); cost = torch.zeros(k, dtype=torch.int32, device="cuda")
        ys = y1.stride(0)
        t_s = timeit(L.uvghip_intra_search_best_batch, [depth, P(y1), ys, P(y1), ys, n, P(blks), k, P(modes), modes.shape[0], P(best), P(cost), None], 5)
        t_p = timeit(L.uvghip_intra_pred_plane_batch, [depth, P(y1), ys, n, P(blks), k, P(best), P(pred), ys])
        t_f = timeit(L.uvghip_tu_forward_batch, [depth, 0, 0, 0, 0, n, n, 0, P(y1), ys, P(pred), ys, P(tus), k, P(coef)])
        t_i = timeit(L.uvghip_tu_inverse_batch, [depth, 0, 0, 0, 0, n, n, 0, P(coef), P(pred), ys, P(rec), ys, P(tus), k])
        d = F if k != cnt else 1
        print(f"n={n:2d} {tag:8s} per picture: search {t_s/d:7.1f} us  pred {t_p/d:6.1f}  tu_forward {t_f/d:6.1f}  tu_inverse {t_i/d:6.1f}")
