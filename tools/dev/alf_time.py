"""Developer probe: time uvghip_alf_stats_batch (luma) on a 1080p frame."""
import sys, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout
y0, _, _ = layout.synthetic_yuv420(1920, 1080, 0, 8)
y1, _, _ = layout.synthetic_yuv420(1920, 1080, 1, 8)
Y0, Y1 = torch.from_numpy(y0).cuda(), torch.from_numpy(y1).cuda()
rects = api.make_rects(layout.ctu_rects(1920, 1080))
cls = api.alf_classify_frame(Y1, 1920, 1080)
for _ in range(2): api.alf_stats_batch(Y0, Y1, rects, cls)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): api.alf_stats_batch(Y0, Y1, rects, cls)
e1.record(); torch.cuda.synchronize()
print(f'alf_stats luma 1080p: {e0.elapsed_time(e1) / 10 * 1000:.1f} us')
