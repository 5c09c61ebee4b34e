"""Developer probe: time uvghip_sad_surface (16x16 blocks, range 8) on a 1080p frame at both bit depths."""
import sys, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout
for depth in (8, 10):
    y0, _, _ = layout.synthetic_yuv420(1920, 1080, 0, depth)
    y1, _, _ = layout.synthetic_yuv420(1920, 1080, 1, depth)
    Y0, Y1 = torch.from_numpy(y0).cuda(), torch.from_numpy(y1).cuda()
    for _ in range(2): api.sad_surface(Y0, Y1, 1920, 1072, 16, 16, 8)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): api.sad_surface(Y0, Y1, 1920, 1072, 16, 16, 8)
    e1.record(); torch.cuda.synchronize()
    print(f'sad_surface {depth}-bit 16x16 r8 1080p: {e0.elapsed_time(e1) / 10 * 1000:.1f} us')
