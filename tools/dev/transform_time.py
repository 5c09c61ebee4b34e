"""Developer probe: uvghip_transform_batch over 2 Mi coefficients per shape / direction."""
import sys, torch
sys.path.insert(0, '.')
from uvg266_amd import api
for n in (4, 8, 16, 32):
    nb = (1920 // n) * (1080 // n)
    b = torch.randint(-255, 256, (nb, n, n), dtype=torch.int16, device='cuda')
    for name, f in (('fwd DCT2', lambda: api.transform_batch(b, 8, False)), ('inv DST7', lambda: api.transform_batch(b, 8, True, 2, 2))):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        print(f'transform {name} {n}x{n}: {e0.elapsed_time(e1) / 10 * 1000:.1f} us')
