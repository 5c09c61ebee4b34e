"""Developer probe: time uvghip_intra_search_batch on a 1080p frame with a chosen candidate list.
usage: search_mode_cost.py [n] [mode|all67] [count] [reps].  Not part of the product or the bench."""
import sys, torch
sys.path.insert(0, '.')
from uvg266_amd import api, layout

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    what = sys.argv[2] if len(sys.argv) > 2 else 'all67'
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    if what == 'all67':
        modes = list(range(67))
    elif what == 'pos':                      # non-negative angles: windows read from the block's pair rows
        modes = [m for m in range(2, 67) if m <= 18 or m >= 50]
    elif what == 'neg':                      # negative angles: windows read from the per-wave strips
        modes = list(range(19, 50))
    else:
        modes = [int(what)] * count
    dev = torch.device('cuda:0')
    y, _, _ = layout.synthetic_yuv420(1920, 1080, 0, 8)
    Y = torch.from_numpy(y).to(dev)
    blks = api.make_intra_blocks(layout.intra_availability(layout.block_grid(1920, 1080, n), n, 1920, 1080), dev)
    m = torch.tensor(modes, dtype=torch.int8, device=dev)
    for _ in range(2):
        api.intra_search_batch(Y, Y, blks, n, m)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        api.intra_search_batch(Y, Y, blks, n, m)
    e1.record(); torch.cuda.synchronize()
    print(f'n={n:2d} {what:8s} x{len(modes):3d} {e0.elapsed_time(e1) / reps * 1000:8.1f} us')
main()
