"""Seeded synthetic frame pairs for parity tests and the bench.

frame1 = band-limited noise (uniform 0..255 -> three 5x5 box blurs -> rescaled to 0..255);
frame2 = frame1 translated by a piecewise-constant integer motion field (tiles x tiles motions
drawn from [-max_motion, max_motion]^2) plus +-2 uniform noise.  Pure numpy, deterministic
for a given (width, height, seed).
"""
import numpy as np


def _box5(a):
    pad = np.pad(a, 2, mode="edge")
    c = np.cumsum(pad, axis=0, dtype=np.float64)
    c = np.vstack([np.zeros((1, c.shape[1])), c])
    v = c[5:] - c[:-5]
    c = np.cumsum(v, axis=1, dtype=np.float64)
    c = np.hstack([np.zeros((c.shape[0], 1)), c])
    return (c[:, 5:] - c[:, :-5]) / 25.0


def synth_pair(width, height, seed, max_motion=24, tiles=4, noise=2):
    """Return (frame1, frame2, motion) with frames uint8 (H, W) and motion int32 (H, W, 2) = (dx, dy)
    such that frame2(y + dy, x + dx) ~= frame1(y, x)."""
    rng = np.random.default_rng(seed)
    big_h, big_w = height + 2 * max_motion, width + 2 * max_motion
    base = rng.integers(0, 256, size=(big_h, big_w)).astype(np.float64)
    for _ in range(3):
        base = _box5(base)
    base -= base.min()
    base *= 255.0 / max(base.max(), 1e-9)
    base = np.rint(base).astype(np.uint8)
    frame1 = base[max_motion:max_motion + height, max_motion:max_motion + width].copy()

    rng2 = np.random.default_rng(seed + 1)
    mv = rng2.integers(-max_motion, max_motion + 1, size=(tiles, tiles, 2))
    ty = np.minimum(np.arange(height) * tiles // height, tiles - 1)
    tx = np.minimum(np.arange(width) * tiles // width, tiles - 1)
    motion = mv[ty[:, None], tx[None, :]].astype(np.int32)          # (H, W, 2) = (dx, dy)
    # frame2(p) = frame1(p - d(p)): each tile of frame2 shows frame1 content moved by +d
    ys, xs = np.mgrid[0:height, 0:width]
    sy = ys - motion[..., 1] + max_motion
    sx = xs - motion[..., 0] + max_motion
    frame2 = base[sy, sx].astype(np.int16)
    if noise:
        frame2 = frame2 + rng2.integers(-noise, noise + 1, size=frame2.shape)
    frame2 = np.clip(frame2, 0, 255).astype(np.uint8)
    return frame1, frame2, motion


def warp_pair_from_flow(flow, seed=4711):
    """A frame pair whose true flow is `flow` ((H, W, 2) float32, Middlebury convention, unknown pixels
    > 1e9): frame2 is a seeded texture, frame1(x) = frame2(x + flow(x)) by bilinear sampling.  Stands in
    for the Middlebury frames, which the reference does not ship (only its ground-truth .flo files)."""
    h, w = flow.shape[:2]
    m = 32
    tex, _, _ = synth_pair(w + 2 * m, h + 2 * m, seed, max_motion=0, noise=0)
    frame2 = tex[m:m + h, m:m + w].copy()
    f = np.where(np.abs(flow) > 1e9, 0.0, flow).astype(np.float64)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = np.clip(xs + f[..., 0] + m, 0, w + 2 * m - 2)
    sy = np.clip(ys + f[..., 1] + m, 0, h + 2 * m - 2)
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0
    t = tex.astype(np.float64)
    frame1 = ((1 - fy) * ((1 - fx) * t[y0, x0] + fx * t[y0, x0 + 1]) +
              fy * ((1 - fx) * t[y0 + 1, x0] + fx * t[y0 + 1, x0 + 1]))
    return np.clip(np.rint(frame1), 0, 255).astype(np.uint8), frame2
