"""Developer probe: ALF classification and 7x7 filter on a 1080p frame."""
import sys, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout
y1, _, _ = layout.synthetic_yuv420(1920, 1080, 1, 8)
Y1 = torch.from_numpy(y1).cuda()
rects_np = layout.ctu_rects(1920, 1080); rects = api.make_rects(rects_np)
cls = api.alf_classify_frame(Y1, 1920, 1080)
coefs = torch.randint(-8, 9, (1, 25, 13), dtype=torch.int16, device='cuda'); coefs[:, :, 12] = 0
clips = torch.full((1, 25, 13), 255, dtype=torch.int16, device='cuda')
sidx = torch.zeros(len(rects_np), dtype=torch.int32, device='cuda'); out = torch.zeros_like(Y1)
for name, f in (('alf_classify', lambda: api.alf_classify_frame(Y1, 1920, 1080)), ('alf_filter 7x7', lambda: api.alf_filter_batch(Y1, out, rects, sidx, coefs, clips, cls))):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 10 * 1000:.1f} us')
