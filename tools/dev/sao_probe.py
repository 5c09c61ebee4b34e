"""Developer probe: sao_apply / sao_stats / deblock timings on a 1080p frame for a few parameter variants."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
from uvg266_amd import api, layout

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1000

dev = torch.device('cuda:0')
W, H = 1920, 1080
y, u, v = layout.synthetic_yuv420(W, H, 0, 8)
Y = torch.from_numpy(y).to(dev); out = torch.zeros_like(Y)
rects_np = layout.ctu_rects(W, H)
rects = api.make_rects(rects_np, dev)
n = len(rects_np)
for name, typ, cls in (('off', 0, 0), ('band', 1, 0), ('eo0', 2, 0), ('eo1', 2, 1), ('eo2', 2, 2), ('eo3', 2, 3)):
    p = np.zeros((n, 8), np.int32); p[:, 0] = typ; p[:, 1] = cls; p[:, 2] = 10; p[:, 3:] = [0, 2, 1, -1, -2]
    P = torch.from_numpy(p).to(dev)
    print('sao_apply', name, round(t(lambda: api.sao_apply_batch(Y, out, rects, P)), 1), 'us')
print('sao_stats', round(t(lambda: api.sao_stats_batch(Y, out, rects)), 1), 'us')
print('empty-ish kernel (sao_edge_offsets)', round(t(lambda: api.sao_edge_offsets_batch(torch.zeros((n, 4, 2, 5), dtype=torch.int32, device=dev))), 1), 'us')
# --- rotation / chain experiment
frames = []
for k in range(4):
    yy, _, _ = layout.synthetic_yuv420(W, H, k, 8)
    frames.append((torch.from_numpy(yy).to(dev), torch.zeros((H, W), dtype=torch.uint8, device=dev)))
p = np.zeros((n, 8), np.int32); p[:, 0] = 2; p[:, 1] = np.arange(n) % 4; p[:, 3:] = [0, 2, 1, -1, -2]
P = torch.from_numpy(p).to(dev)
state = {'i': 0}
def rot():
    a, b = frames[state['i'] % 4]; state['i'] += 1
    api.sao_apply_batch(a, b, rects, P)
print('sao_apply rotating 4 frames, mixed classes', round(t(rot, 20), 1), 'us')
big = torch.zeros((64 * 1024 * 1024,), dtype=torch.uint8, device=dev)
def rot_flush():
    big.add_(1)                      # 64 MB read+write between calls: evicts L2
    rot()
print('  ... with a 64 MB elementwise op in between (per pair)', round(t(rot_flush, 20), 1), 'us')
print('  64 MB elementwise op alone', round(t(lambda: big.add_(1), 20), 1), 'us')
def rot_copy():
    a, b = frames[state['i'] % 4]; state['i'] += 1
    b.copy_(a)
print('torch copy_ rotating 4 frames', round(t(rot_copy, 20), 1), 'us')
p1 = p.copy(); p1[:, 0] = 1; P1 = torch.from_numpy(p1).to(dev)
def rot_band():
    a, b = frames[state['i'] % 4]; state['i'] += 1
    api.sao_apply_batch(a, b, rects, P1)
print('sao_apply band rotating', round(t(rot_band, 20), 1), 'us')
p2 = p.copy(); p2[:, 1] = 0; P2 = torch.from_numpy(p2).to(dev)
def rot_eo0():
    a, b = frames[state['i'] % 4]; state['i'] += 1
    api.sao_apply_batch(a, b, rects, P2)
print('sao_apply eo0 rotating', round(t(rot_eo0, 20), 1), 'us')
def same_mixed():
    a, b = frames[0]
    api.sao_apply_batch(a, b, rects, P)
print('sao_apply mixed classes, same frame', round(t(same_mixed, 20), 1), 'us')
