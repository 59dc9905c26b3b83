#!/usr/bin/env python3
"""Soak: many random configurations (the generator of tests/test_gpu_parity.py) through the whole pyramid, each under a
few scheduling settings, every field against the CPU oracle's.  Development aid for changes to the regulariser's schedule
(round 4: the SAD memo, forwarding, the fix-up list split).   python scripts/soak_random.py [cases] [first_seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                               # noqa: E402
import blockbasedmotionestimation_amd as bbme                    # noqa: E402
from oracle import bbme_oracle as oracle                         # noqa: E402
from test_gpu_parity import _random_case                          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"     # 720p .. 1080p frames: thousands of solver waves at once


def _big_case(rng):
    levels = int(rng.integers(2, 4))
    b = int(rng.choice([8, 16, 32]))
    blocks = [b] * levels
    m = b << (levels - 1)
    w = int(rng.integers(1280 // m, 1920 // m + 1)) * m
    h = int(rng.integers(704 // m, 1088 // m + 1)) * m
    search = [b + 2 * int(rng.integers(2, 17))] * levels
    kind = int(rng.integers(0, 3))
    if kind == 0:
        from blockbasedmotionestimation_amd.synth import synth_pair
        f1, f2, _ = synth_pair(w, h, int(rng.integers(1 << 30)), max_motion=int(rng.integers(4, 24)))
    elif kind == 1:
        f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f2 = np.roll(f1, (int(rng.integers(-9, 10)), int(rng.integers(-9, 10))), axis=(0, 1))
        f2[h // 3:, w // 2:] = rng.integers(0, 256, (h - h // 3, w - w // 2), dtype=np.uint8)
    else:
        f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f2 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return f1, f2, search, blocks
SETTINGS = [{}, {"BBME_MEMO_MIN_B": "8", "BBME_MEMO_FORWARD": "1"}, {"BBME_MEMO_FORWARD": "1", "BBME_SPEC_MIN_GABS": "0"},
            {"BBME_MEMO": "0", "BBME_SPEC_MIN_GABS": "0", "BBME_LIST_SPLIT": "0"},
            {"BBME_SOLVE_WGS": "1", "BBME_SOLVE_WAVES": "1", "BBME_MEMO_MIN_B": "8"},
            {"BBME_PASS1_STRIP": "1", "BBME_PASS1_LANES_MAX": "0"},
            {"BBME_PASS1_LANES_MAX": "0", "BBME_RELAX_STEPS": "1"},            # lazy pass 1 in front of a relaxation launch, every sweep
            {"BBME_PASS1_LAZY": "0", "BBME_PASS1_LANES_MAX": "0", "BBME_RELAX_STEPS": "2"},
            {"BBME_SOLVE_WGS": "8", "BBME_WIDE_THRESHOLD": "100000"},           # long queues: every round hands blocks to idle siblings
            {"BBME_SOLVE_WGS": "16", "BBME_SOLVE_WAVES": "2", "BBME_WIDE_THRESHOLD": "1", "BBME_MEMO_MIN_B": "8"},
            {"BBME_SOLVE_SHARE": "0"}]
ran = bad = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    f1, f2, search, blocks = _big_case(rng) if BIG else _random_case(rng)
    L = len(blocks)
    try:
        omf = oracle.OracleMF(f1, f2, search, blocks, use_cache=False)
    except ValueError:
        continue
    if any((omf.level_shape(l)[0] // blocks[l] < 2) or (omf.level_shape(l)[1] // blocks[l] < 2) for l in range(L)):
        omf.close()
        continue
    exp = omf.calc_motion_block_matching()
    for env in SETTINGS:
        os.environ.update(env)
        try:
            mf = bbme.MF(f1, f2, search, blocks, L)
        finally:
            for k in env:
                del os.environ[k]
        for lvl in range(L):
            mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
        for rep in range(2):
            got = mf.calcMotionBlockMatching()
            if not np.array_equal(got, exp):
                bad += 1
                print("MISMATCH seed %d %s rep %d: %s search %s blocks %s: %d values differ" %
                      (seed, env, rep, f1.shape, search, blocks, int((got != exp).sum())), flush=True)
        mf.close()
    omf.close()
    ran += 1
    if ran % (2 if BIG else 25) == 0:
        print("%d cases, %d mismatches, %.0f s" % (ran, bad, time.time() - t0), flush=True)
print("soak done: %d cases x %d settings x 2 runs, %d mismatches" % (ran, len(SETTINGS), bad))
sys.exit(1 if bad else 0)
