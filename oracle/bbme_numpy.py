"""Second, independently structured restatement of the hot path in numpy / pure Python.

TEST INFRASTRUCTURE ONLY (small cases; pure-Python loops).  It deliberately does NOT follow the
C oracle's structure, so that agreement between the two means something:
  * the spiral tie-break uses the CLOSED-FORM rank of SURVEY.md appendix A.4 instead of walking
    the loop of motion_framework.cpp:326-411;
  * the regulariser's neighbour list is the single filtered order C,L,R,DR,UL,UR,U,D,DL
    instead of the nine branches of :439-522;
  * energies are exact Python integers instead of float32 (:607);
  * MVs live in compact per-block integer grids instead of a dense float field, and
    divide_blocks / copyMVs / copy_to_all_pixels are index arithmetic.
PARITY UNPINNED with respect to the reference binary (see oracle/bbme_oracle.h).
"""
import numpy as np

_ORDER = [(0, 0), (0, -1), (0, 1), (1, 1), (-1, -1), (-1, 1), (-1, 0), (1, 0), (1, -1)]


def spiral_rank(dx, dy):
    """Visit index of offset (dx, dy) in the reference's spiral (closed form)."""
    r = max(abs(dx), abs(dy))
    if r == 0:
        return 0
    base = lambda q: 2 * (q - 1) * (2 * q - 1)          # noqa: E731
    m = 2 * r - 1
    if dy == -r and dx > -r:
        return base(r + 1) + (dx + r)
    if dx == r:
        return base(r) + m + (dy + r - 1)
    if dy == r:
        return base(r) + 2 * m + (r - dx)
    return base(r) + 3 * m + 1 + (r - dy)


def _sad(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).sum())


def search_level(img1, img2, B, search_size, pred):
    """pred: (rows, cols, 2) integer prediction per block.  Returns the MV grid."""
    H, W = img1.shape
    R = max(0, (search_size - B) >> 1)
    rows, cols = H // B, W // B
    out = np.zeros((rows, cols, 2), np.int64)
    for r in range(rows):
        for c in range(cols):
            i, j = r * B, c * B
            px, py = j + int(pred[r, c, 0]), i + int(pred[r, c, 1])
            if px < 0 or py < 0 or px + B > W or py + B > H:
                continue                                    # zero MV, no search
            cur = img1[i:i + B, j:j + B]
            best = None
            for dy in range(-R, R + 1):
                for dx in range(-R, R + 1):
                    x, y = px + dx, py + dy
                    if x < 0 or y < 0 or x + B > W or y + B > H:
                        continue
                    key = (_sad(cur, img2[y:y + B, x:x + B]), spiral_rank(dx, dy))
                    if best is None or key < best[0]:
                        best = (key, x, y)
            out[r, c] = (best[1] - j, best[2] - i)
    return out


def sweep(img1, img2, grid, b, lam_times_mult):
    """One in-place raster sweep at block size b on an integer MV grid (rows, cols, 2)."""
    H, W = img1.shape
    rows, cols = grid.shape[:2]
    for r in range(rows):
        for c in range(cols):
            cands = [tuple(int(v) for v in grid[r + dr, c + dc]) for dr, dc in _ORDER
                     if 0 <= r + dr < rows and 0 <= c + dc < cols]
            best_e, best_k = None, 0
            for k, (u, v) in enumerate(cands):
                x, y = c * b + u, r * b + v
                if x < 0 or x > W - b or y < 0 or y > H - b:
                    e = None                                # FLT_MAX
                else:
                    smooth = sum(abs(uu - u) + abs(vv - v) for uu, vv in cands)
                    e = _sad(img1[r * b:(r + 1) * b, c * b:(c + 1) * b], img2[y:y + b, x:x + b]) \
                        + lam_times_mult * smooth
                if k == 0:
                    best_e, best_k = e, 0
                elif e is not None and (best_e is None or e < best_e):
                    best_e, best_k = e, k
            grid[r, c] = cands[best_k]
    return grid


def run(planes1, planes2, search_size, block_size, on_stage=None):
    """Whole pyramid on ready-made level planes.  Returns the dense float32 flow of level 0."""
    L = len(block_size)
    final2 = [None] * L
    for lvl in range(L - 1, -1, -1):
        i1, i2 = planes1[lvl], planes2[lvl]
        H, W = i1.shape
        B = block_size[lvl]
        rows, cols = H // B, W // B
        pred = np.zeros((rows, cols, 2), np.int64)
        if lvl != L - 1:
            Bc = block_size[lvl + 1]
            for r in range(rows):
                for c in range(cols):
                    ci, cj = (r * B) // (2 * Bc) * Bc, (c * B) // (2 * Bc) * Bc
                    pred[r, c] = 2 * final2[lvl + 1][ci // 2, cj // 2]
        grid = search_level(i1, i2, B, search_size[lvl], pred)
        if on_stage:
            on_stage("search", lvl, B, grid.copy())
        b, lam = B, B // 2
        while b > 1:
            for mult in (1, 2):
                sweep(i1, i2, grid, b, lam * mult)
                if on_stage:
                    on_stage("sweep%d" % mult, lvl, b, grid.copy())
            if b > 2:
                grid = np.repeat(np.repeat(grid, 2, 0), 2, 1)     # divide_blocks
            b >>= 1
            lam *= 2
        final2[lvl] = grid
    return np.repeat(np.repeat(final2[0], 2, 0), 2, 1).astype(np.float32)
