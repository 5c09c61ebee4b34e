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


SCU_DTYPE = np.dtype([("luma_edges", "u1"), ("chroma_edges", "u1"), ("type", "u1"), ("cbf", "u1"), ("qp", "i1"),
                      ("log2_width", "u1"), ("log2_height", "u1"), ("log2_chroma_width", "u1"), ("log2_chroma_height", "u1"),
                      ("isp_mode", "u1"), ("mv_dir", "u1"), ("reserved", "u1"), ("ref_id", "<i2", (2,)),
                      ("mv", "<i4", (2, 2))])   # uvghip_scu_t, include/uvg266_hip.h


def quadtree_scu_table(pic_w, pic_h, seed=0, inter=False, qp=None, split_prob=None):
    """A seeded random quad-tree partition of every CTU into square CUs (4..64) written as the per-4x4
    side-information table deblocking needs (uvghip_scu_t), with edge flags placed like the reference's
    mark_deblocking (src/search.c:1075-1110: CU edges, plus the 32-sample TU split of 64-wide CUs)."""
    rng = np.random.default_rng(seed)
    ts, th = ((pic_w + CTU - 1) // CTU) * 16, ((pic_h + CTU - 1) // CTU) * 16
    tab = np.zeros((th, ts), SCU_DTYPE)
    prob = split_prob or {64: 0.85, 32: 0.6, 16: 0.5, 8: 0.4}

    def split(x, y, size):
        if x >= pic_w or y >= pic_h:
            return
        must = x + size > pic_w or y + size > pic_h
        if size > 4 and (must or rng.random() < prob[size]):
            h = size // 2
            for dx, dy in ((0, 0), (h, 0), (0, h), (h, h)):
                split(x + dx, y + dy, h)
            return
        lg = size.bit_length() - 1
        intra = (not inter) or rng.random() < 0.3
        cu = tab[y // 4:(y + size) // 4, x // 4:(x + size) // 4]
        cu["type"] = 1 if intra else 2
        cu["cbf"] = rng.integers(0, 8)
        cu["qp"] = rng.integers(20, 45) if qp is None else qp
        cu["log2_width"] = cu["log2_height"] = lg
        cu["log2_chroma_width"] = cu["log2_chroma_height"] = max(lg - 1, 2)
        if not intra:
            cu["mv_dir"] = rng.integers(1, 4)
            cu["mv"] = rng.integers(-20, 21, (2, 2))
            cu["ref_id"] = rng.integers(0, 3, 2)
        ex = np.zeros((size // 4, size // 4), np.uint8)
        xs = np.arange(x, x + size, 4)
        ys = np.arange(y, y + size, 4)
        ver = ((xs - x) % 32 == 0) if x > 0 else ((xs == 32) & (size == 64))
        hor = ((ys - y) % 32 == 0) if y > 0 else ((ys == 32) & (size == 64))
        ex |= ver[None, :].astype(np.uint8)
        ex |= (hor[:, None].astype(np.uint8) << 1)
        cu["luma_edges"] = ex
        cu["chroma_edges"] = ex
    for cy in range(0, pic_h, CTU):
        for cx in range(0, pic_w, CTU):
            split(cx, cy, CTU)
    return tab


def ctu_rects(pic_w, pic_h, ctu=CTU):
    """(n,4) int32 rows (x, y, w, h): the CTU grid clipped to the picture = uvghip_rect_t."""
    return np.array([[x, y, min(ctu, pic_w - x), min(ctu, pic_h - y)]
                     for y in range(0, pic_h, ctu) for x in range(0, pic_w, ctu)], np.int32)


def frames_of_rank(rank, world, n_frames):
    """Frames a rank owns when a sequence is sharded over `world` ranks (all-intra: frames are
    independent units): frame t belongs to rank t % world."""
    return list(range(rank, n_frames, world))


def coding_order(xy, n):
    """Indices that put uniform n x n blocks into coding order: CTUs in raster order, quad-tree z-order inside a CTU."""
    xy = np.asarray(xy, np.int64)
    x, y = xy[:, 0], xy[:, 1]
    z = _zindex((x % CTU) // 4, (y % CTU) // 4)
    return np.lexsort((z, x // CTU, y // CTU))


def dependency_levels(blks, n):
    """Wavefront levels of closed-loop intra coding for uniform n x n blocks: level[b] = 1 + max(level of every block whose
    reconstruction b's reference samples come from) -- the row above from x - 1 to x + avail_top - 1 and the column to the left
    down to y + avail_left - 1 (uvg_intra_build_reference, src/intra.c:1252-1318).  `blks`: (k,4) rows of intra_availability.
    Blocks of one level are independent; levels must run in order.  This is the reference's WPP dependency
    (encoderstate.c:1085-1189: a CTU waits for its left and above-right neighbours) refined to CU granularity."""
    blks = np.asarray(blks, np.int64)
    index = {(int(bx), int(by)): i for i, (bx, by) in enumerate(zip(blks[:, 0] // n, blks[:, 1] // n))}
    level = np.full(len(blks), -1, np.int32)
    for i in coding_order(blks[:, :2], n):
        x, y, at, al = (int(v) for v in blks[i])
        bx, by = x // n, y // n
        deps = []
        if y > 0:
            deps += [(k, by - 1) for k in range(bx - (1 if x > 0 else 0), (x + max(at, 1) - 1) // n + 1)]
        if x > 0:
            deps += [(bx - 1, k) for k in range(by, (y + max(al, 1) - 1) // n + 1)]
        lv = -1
        for d in deps:
            j = index.get(d)
            if j is not None:
                assert level[j] >= 0, "reference to a block that is coded later"
                lv = max(lv, int(level[j]))
        level[i] = lv + 1
    return level
