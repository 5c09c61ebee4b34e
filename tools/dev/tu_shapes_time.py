"""Developer probe: fused TU round trip over a 1080p frame for square, rectangular and zero-out shapes."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout
y0, _, _ = layout.synthetic_yuv420(1920, 1080, 0, 8)
y1, _, _ = layout.synthetic_yuv420(1920, 1080, 1, 8)
Y0, Y1 = torch.from_numpy(y0).cuda(), torch.from_numpy(y1).cuda()
rec = torch.zeros_like(Y0)
for (w, h, th, tv, sw, sh) in [(8, 8, 0, 0, 0, 0), (16, 16, 0, 0, 0, 0), (32, 32, 0, 0, 0, 0), (16, 8, 0, 0, 0, 0), (8, 16, 0, 0, 0, 0),
                               (32, 16, 0, 0, 0, 0), (4, 8, 0, 0, 0, 0), (32, 32, 1, 1, 16, 16), (16, 16, 1, 2, 0, 0), (8, 8, 2, 1, 0, 0)]:
    xs, ys = np.meshgrid(np.arange(0, 1920 - w + 1, w), np.arange(0, 1080 - h + 1, h))
    tus = api.make_tus(np.stack([xs.ravel(), ys.ravel()], 1))
    f = lambda: api.tu_roundtrip_batch(Y0, Y1, rec, tus, w, h, 22, True, th, tv, sw, sh)
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'tu_roundtrip {w}x{h} types ({th},{tv}) skip ({sw},{sh}): {e0.elapsed_time(e1) / 10 * 1000:.1f} us')
