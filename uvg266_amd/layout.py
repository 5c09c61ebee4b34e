"""Host-side tables describing how a picture is cut into blocks (CTU grid, quad-tree
z-order) -- the information the reference keeps implicitly in lcu_t / cu_array_t and
that the batched kernels take as descriptor arrays."""
import numpy as np

CTU = 64  # LCU_WIDTH, src/global.h:185


def _zindex(sx, sy):
    z = np.zeros_like(sx)
    for b in range(4):
        z |= ((sx >> b) & 1) << (2 * b)
        z |= ((sy >> b) & 1) << (2 * b + 1)
    return z


def block_grid(pic_w, pic_h, n):
    """All n x n blocks that lie completely inside the picture, raster order -> (k,2) int32 (x, y)."""
    xs, ys = np.meshgrid(np.arange(0, pic_w - n + 1, n), np.arange(0, pic_h - n + 1, n))
    return np.stack([xs.ravel(), ys.ravel()], 1).astype(np.int32)


def intra_availability(xy, n, pic_w, pic_h):
    """px_available_top / px_available_left (src/intra.c:1252-1318 without the WPP clamp) for n x n
    blocks coded in quad-tree z-order inside 64x64 CTUs that are coded in raster order.
    Returns (k,4) int32 rows (x, y, avail_top, avail_left) = uvghip_intra_blk_t."""
    xy = np.asarray(xy, np.int64)
    x, y = xy[:, 0], xy[:, 1]
    lx, ly = x % CTU, y % CTU
    zc = _zindex(lx // 4, ly // 4)

    def coded(px, py):
        inside = (px >= 0) & (py >= 0) & (px < pic_w) & (py < pic_h)
        cx, cy = (px // CTU) * CTU, (py // CTU) * CTU
        ox, oy = x - lx, y - ly
        earlier_ctu = (cy < oy) | ((cy == oy) & (cx < ox))
        # only the left, above-left, above and above-right CTUs are ever referenced
        near = (cx <= ox + CTU) | (cy < oy)
        same = (cx == ox) & (cy == oy)
        zin = _zindex(((px - cx) // 4).clip(0, 15), ((py - cy) // 4).clip(0, 15))
        return inside & ((earlier_ctu & near) | (same & (zin < zc)))

    at = np.zeros(len(xy), np.int64)
    al = np.zeros(len(xy), np.int64)
    alive_t = np.ones(len(xy), bool)
    alive_l = np.ones(len(xy), bool)
    for k in range(0, 2 * n, 4):
        alive_t &= coded(x + k, y - 1)
        at += 4 * alive_t
        alive_l &= coded(x - 1, y + k)
        al += 4 * alive_l
    at = np.minimum(np.minimum(at, 2 * n), pic_w - x)
    al = np.minimum(np.minimum(al, 2 * n), pic_h - y)
    return np.stack([x, y, at, al], 1).astype(np.int32)


def synthetic_yuv420(w, h, t, depth=8, seed=1234):
    """The synthetic picture generator of SURVEY.md section 8(d): sin/cos field + Gaussian noise +
    a bright 64x64 square moving (+3,+2) px/frame.  Returns (Y, U, V) numpy planes."""
    rng = np.random.default_rng(seed + t)
    s = 1 if depth == 8 else 4
    mid, a1, a2, a3, sigma = 128 * s, 60 * s, 50 * s, 30 * s, 4 * s
    yy, xx = np.mgrid[0:h, 0:w]
    Y = mid + a1 * np.sin((xx + 3 * t) / 37.0) + a2 * np.cos((yy - 2 * t) / 23.0) + rng.normal(0, sigma, (h, w))
    sx, sy = (100 + 3 * t) % max(1, w - 64), (60 + 2 * t) % max(1, h - 64)
    Y[sy:sy + 64, sx:sx + 64] += 60 * s
    cy, cx = np.mgrid[0:h // 2, 0:w // 2]
    U = mid + a3 * np.sin((cx + t) / 50.0) + 0 * cy
    V = mid + a3 * np.cos((cy + t) / 40.0) + 0 * cx
    dt = np.uint8 if depth == 8 else np.uint16
    mx = (1 << depth) - 1
    return tuple(np.clip(np.rint(p), 0, mx).astype(dt) for p in (Y, U, V))
