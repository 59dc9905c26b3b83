"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact motion vectors (integer work) at every stage of the
reference's schedule -- after the search of each level and after every regulariser sweep --
and bit-exact dense float flow at the end."""
import numpy as np
import pytest

from helpers import compare_stagewise, gpu_schedule, oracle_schedule

pytestmark = pytest.mark.gpu


def test_isa_probes(bbme):
    """v_sad_u8 / v_alignbyte_b32 / v_qsad_pk_u16_u8 / v_sad_u16 behave as the kernels assume."""
    import ctypes as C
    from blockbasedmotionestimation_amd import _capi
    mism = (C.c_int * 5)()
    _capi.check(_capi.lib().bbme_selftest_isa(0, mism))
    assert list(mism) == [0] * 5, "sad_u8, alignbyte, qsad_pk_u16_u8, sad_u16, unaligned loads: %s" % list(mism)


def test_device_probes(bbme):
    """The measurements bench.py quotes come from these entry points: the XCD round-robin check behind the XCD-aware block
    orders (8 XCDs, no workgroup off its residue class) and the two search inner loops (both must report a rate)."""
    import ctypes as C
    from blockbasedmotionestimation_amd import _capi
    seen, viol = C.c_int(), C.c_int()
    _capi.check(_capi.lib().bbme_probe_xcd(0, C.byref(seen), C.byref(viol)))
    assert seen.value == 8 and viol.value == 0
    t = (C.c_double * 2)()
    _capi.check(_capi.lib().bbme_probe_search_loops(0, t))
    assert 20 < t[0] < 400 and 20 < t[1] < 400


CASES = [
    # (width, height, search_size[], block_size[], seed, max_motion)
    (320, 208, [30, 30, 30], [16, 16, 16], 1001, 12),          # cfg1-like: B=16, R=7, 3 levels
    (256, 192, [48], [16], 1002, 14),                           # single level, R=16
    (384, 256, [48, 48, 48], [16, 16, 16], 1003, 24),           # cfg2-like, 3 levels
    (512, 384, [80, 80, 80], [16, 16, 16], 1004, 40),           # R=32 (cfg3's search), windows leave the image
    (256, 256, [72, 72], [8, 8], 1005, 20),                     # cfg4-like: B=8, R=32
    (512, 512, [64, 64, 64], [32, 32, 32], 1006, 30),           # the reference's own literals: B=32, search 64
    (320, 256, [24, 40, 30], [8, 16, 8], 1007, 10),             # different block / search per level
    (200, 120, [30, 30], [16, 16], 1008, 6),                    # needs padding in both dimensions
    (256, 128, [17, 21], [16, 16], 1009, 3),                    # odd shift (search-block odd), tiny ranges
    (128, 128, [16], [16], 1010, 0),                            # search_size == block_size: centre only
    (256, 192, [12, 12], [4, 4], 1011, 5),                      # B=4
    (512, 512, [80, 80], [64, 64], 1012, 10),                   # B=64 (generic search, 64-lane regulariser groups)
    (640, 512, [20, 20, 20, 24, 24], [4, 4, 4, 8, 8], 1013, 30),  # five levels, large coarse-to-fine motion
    (256, 128, [8, 12], [16, 16], 1014, 2),                     # search_size < block_size: centre candidate only
    (250, 130, [30], [16], 1015, 5),                            # odd-looking size, padded both ways (256 x 144)
    (1024, 64, [48, 48], [16, 16], 1016, 12),                   # two block rows at the coarse level, very wide
    (64, 1024, [48, 48], [16, 16], 1017, 12),                   # two block columns, very tall
    # wide ranges: the fast kernel's packed (SAD, rank) keys at their limits (B=32: ranks up to 16128 need 14 bits)
    (512, 384, [120], [32], 1018, 40),                          # B=32, R=44
    (512, 384, [122], [32], 1019, 44),                          # B=32, R=45: 8281 candidates > 2^13
    (512, 384, [123], [32], 1020, 44),                          # B=32, odd shift, R=45
    (384, 384, [158], [32], 1021, 60),                          # B=32, R=63 (largest supported)
    (384, 256, [134], [8], 1022, 60),                           # B=8, R=63
    (384, 256, [142], [16], 1023, 60),                          # B=16, R=63
    # ranges beyond the strip kernel's packed keys (R > 63) take the generic kernel, whose window then needs more LDS than a kernel
    # gets by default (r04; the reference takes any search size, motion_framework.cpp:296-422)
    (384, 256, [16 + 2 * 64], [16], 1025, 60),                  # B=16, R=64: the first range past the strip kernel
    (320, 256, [8 + 2 * 100, 8 + 2 * 70], [8, 8], 1026, 70),    # B=8, R=100 over R=70
    (384, 384, [32 + 2 * 127], [32], 1027, 100),                # B=32, R=127 (largest supported): 286-row window, 87 KB of LDS
    # 2 x 2 blocks as a level's own block size (r04; the generic search kernel with the block in one dword): alone, under 4 x 4, and
    # between two levels of larger blocks (copyMVs from a level that is already at 2 x 2 cells when its search ends)
    (128, 96, [10], [2], 1028, 3),
    (160, 128, [12, 20], [2, 4], 1029, 4),
    (192, 128, [14, 10, 24], [4, 2, 8], 1030, 5),
    # the author's second literal set (main_class.cpp:15-17, commented out there) on the 584 x 388 Middlebury geometry:
    # 32 x 32 blocks over 16 x 16 ones (search_prediction's mixed-size path) and an odd shift, 42 - 32 = 10 -> R = 5
    (584, 388, [32, 32, 42], [16, 16, 32], 1024, 10),
]


@pytest.mark.parametrize("w,h,search,block,seed,mm", CASES)
def test_stagewise_parity(bbme, oracle, w, h, search, block, seed, mm):
    f1, f2, _ = bbme.synth_pair(w, h, seed, max_motion=mm)
    compare_stagewise(bbme, oracle, f1, f2, search, block)


@pytest.mark.parametrize("name", ["hotpath_b16_r7_l3", "hotpath_b16_r16_l2", "hotpath_b8_r32_l2",
                                  "hotpath_b32_r16_l2", "hotpath_mixed_l3", "hotpath_ref2_l3", "hotpath_block2_l2",
                                  "variant_raster_b16_r7_l3", "variant_raster_b8_r32_l2", "variant_jacobi_b16_r7_l3"])
def test_golden_fixtures(bbme, name):
    """Committed vectors (tests/golden/*.npz): planes in, every intermediate MV grid and the final
    dense flow out.  No oracle call on this path."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    block = g["block_size"].tolist()
    L = len(block)
    mf = bbme.MF(g["frame1"], g["frame2"], g["search_size"].tolist(), block, L)
    mf.set_search_mode("raster" in name)
    mf.set_regularizer_mode("jacobi" in name)
    assert [mf.padded_width, mf.padded_height, mf.padding_x, mf.padding_y] == g["geometry"].tolist()
    for lvl in range(L):
        mf.set_level_planes(lvl, g["plane1_l%d" % lvl], g["plane2_l%d" % lvl])
    got = []
    flow = gpu_schedule(mf, L, block, lambda kind, lvl, b, mv: got.append((kind, lvl, b, mv)))
    keys = [str(k) for k in g["stages"]]
    assert len(keys) == len(got)
    for key, (kind, lvl, b, mv) in zip(keys, got):
        assert key.endswith("%s_l%d_b%d" % (kind, lvl, b))
        assert np.array_equal(g[key].astype(np.int32), mv), key
    assert np.array_equal(g["flow"], flow)
    mf.close()


def _random_case(rng):
    """A random legal configuration and frame pair (small enough for the oracle to take milliseconds)."""
    levels = int(rng.integers(1, 4))
    blocks = [int(rng.choice([2, 4, 4, 8, 8, 16, 16, 32])) for _ in range(levels)]
    # sizes that need no padding keep the search for a legal size trivial; padding is tested elsewhere; every level's width a
    # multiple of four (the kernels move rows as dwords: only 2 x 2 blocks can ask for less)
    m = int(np.lcm.reduce([b << i for i, b in enumerate(blocks)] + [4 << (levels - 1)]))
    w = m * int(rng.integers(max(2, -(-2 * (blocks[-1] << (levels - 1)) // m)), 6))
    h = m * int(rng.integers(max(2, -(-2 * (blocks[-1] << (levels - 1)) // m)), 5))
    w, h = min(w, 768), min(h, 512)
    w, h = max(m * 2, w // m * m), max(m * 2, h // m * m)
    search = [b + 2 * int(rng.integers(0, 20)) + int(rng.integers(0, 2)) for b in blocks]
    kind = int(rng.integers(0, 5))
    if kind == 0:                       # smooth texture + piecewise motion (the bench's recipe)
        from blockbasedmotionestimation_amd.synth import synth_pair
        f1, f2, _ = synth_pair(w, h, int(rng.integers(1 << 30)), max_motion=int(rng.integers(0, 12)))
    elif kind == 1:                     # white noise, shifted
        f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f2 = np.roll(f1, (int(rng.integers(-9, 10)), int(rng.integers(-9, 10))), axis=(0, 1))
    elif kind == 2:                     # few grey levels: ties everywhere
        f1 = (rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)
        f2 = (rng.integers(0, 3, (h, w)) * 100).astype(np.uint8)
    elif kind == 3:                     # flat regions next to texture
        f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f1[: h // 2, : w // 2] = 50
        f2 = np.roll(f1, 3, axis=1)
        f2[h // 3:, w // 3:] = 200
    else:                               # unrelated frames
        f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        f2 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return f1, f2, search, blocks


@pytest.mark.parametrize("seed", range(40))
def test_random_configurations_twice(bbme, oracle, seed):
    """Random sizes, block sizes, ranges, level counts and image statistics; every case runs twice on the
    GPU (the asynchronous solver's schedule differs from run to run, its result must not) and must equal
    the oracle's field bit for bit."""
    rng = np.random.default_rng(9000 + seed)
    f1, f2, search, blocks = _random_case(rng)
    L = len(blocks)
    try:
        omf = oracle.OracleMF(f1, f2, search, blocks)
    except ValueError:
        pytest.skip("illegal geometry for the reference")
    if any((omf.level_shape(l)[0] // blocks[l] < 2) or (omf.level_shape(l)[1] // blocks[l] < 2) for l in range(L)):
        pytest.skip("degenerate grid (undefined in the reference)")
    exp = omf.calc_motion_block_matching()
    mf = bbme.MF(f1, f2, search, blocks, L)
    for lvl in range(L):
        mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
    a = mf.calcMotionBlockMatching()
    b = mf.calcMotionBlockMatching()
    mf.close()
    assert np.array_equal(a, exp), "first run differs from the oracle: %s %s %s" % (f1.shape, search, blocks)
    assert np.array_equal(b, exp), "second run differs from the oracle"
    # the split of the work between the tile kernel's local fixed point and the solver, and the number of solver
    # waves, are pure scheduling: a solver of one workgroup of one wave must leave the same field
    import os
    # (so is the number of waves that share a macroblock in the search: two by default on levels of <= 1024 blocks --
    # never here, and on every level in the relaxation runs below;
    # and the form of pass 1: the chain form by default on grids of this size, the throughput form here)
    os.environ["BBME_SOLVE_WGS"], os.environ["BBME_SOLVE_WAVES"], os.environ["BBME_SEARCH_SPLIT_BLOCKS"] = "1", "1", "0"
    os.environ["BBME_PASS1_LANES_MAX"], os.environ["BBME_SCAN_FINE_MAX"] = "0", "0"      # ... and 16-flag scan segments
    try:
        mf = bbme.MF(f1, f2, search, blocks, L)
    finally:
        del os.environ["BBME_SOLVE_WGS"], os.environ["BBME_SOLVE_WAVES"], os.environ["BBME_SEARCH_SPLIT_BLOCKS"]
        del os.environ["BBME_PASS1_LANES_MAX"], os.environ["BBME_SCAN_FINE_MAX"]
    for lvl in range(L):
        mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
    c = mf.calcMotionBlockMatching()
    mf.close()
    assert np.array_equal(c, exp), "a one-wave solver changes the field"
    # the tile-resident relaxation launches before the solver (by default only on grids of >= 300 000 blocks at b = 2)
    # forced into every sweep: any number of them must leave the result untouched
    for steps in ("1", "3"):
        # ... and with the speculative search forced onto every level (by default only levels with >= 8 G abs-diffs of
        # search work): every block's search starts early from a provisional prediction, the blocks whose prediction
        # changed are searched again, and the result must still be the search from the final prediction
        os.environ["BBME_RELAX_STEPS"], os.environ["BBME_SEARCH_SPLIT_BLOCKS"] = steps, "100000000"
        os.environ["BBME_SPEC_MIN_GABS"] = "0"
        try:
            mf = bbme.MF(f1, f2, search, blocks, L)
        finally:
            del os.environ["BBME_RELAX_STEPS"], os.environ["BBME_SEARCH_SPLIT_BLOCKS"], os.environ["BBME_SPEC_MIN_GABS"]
        for lvl in range(L):
            mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
        c = mf.calcMotionBlockMatching()
        mf.close()
        assert np.array_equal(c, exp), "%s relaxation steps per sweep change the field" % steps
    # r04: a solver wave hands the blocks it cannot take in a round to idle waves of its workgroup (LDS mailboxes).  With one
    # workgroup per XCD every queue is long and every round donates: chain rounds only (mailboxes filled to their cap), and
    # two-wave workgroups whose rounds all take the throughput form; and with the hand-over switched off
    for env in ({"BBME_SOLVE_WGS": "8", "BBME_WIDE_THRESHOLD": "100000"},
                {"BBME_SOLVE_WGS": "8", "BBME_SOLVE_WAVES": "2", "BBME_WIDE_THRESHOLD": "1"},
                {"BBME_SOLVE_SHARE": "0"}):
        os.environ.update(env)
        try:
            mf = bbme.MF(f1, f2, search, blocks, L)
        finally:
            for key in env:
                del os.environ[key]
        for lvl in range(L):
            mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
        c = mf.calcMotionBlockMatching()
        d = mf.calcMotionBlockMatching()
        mf.close()
        assert np.array_equal(c, exp) and np.array_equal(d, exp), "solver setting %s changes the field" % env


def test_non_convergence_is_an_error(bbme):
    """The regulariser's waves leave at a round cap instead of spinning for ever.  A sweep that hit the cap has not
    reached the reference's field: every call that hands out a result must fail loudly (BBME_ERR_STATE), and the
    context must work again afterwards.  The cap is forced to one round with a test knob."""
    import os
    from blockbasedmotionestimation_amd import _capi
    f1, f2, _ = bbme.synth_pair(512, 384, 77, max_motion=20)
    search, block = [48, 48], [16, 16]
    ref = bbme.MF(f1, f2, search, block, 2)
    expect = ref.calcMotionBlockMatching()
    ref.close()
    os.environ["BBME_TEST_ROUND_CAP"] = "1"
    try:
        mf = bbme.MF(f1, f2, search, block, 2)
    finally:
        del os.environ["BBME_TEST_ROUND_CAP"]
    for call in (mf.calcMotionBlockMatching, lambda: (mf.estimate_async(), mf.synchronize()),
                 lambda: (mf.estimate_async(), mf.get_cells())):
        with pytest.raises(_capi.BbmeError) as err:
            call()
        assert err.value.status == _capi.ERR_STATE and "converg" in err.value.message
    mf.close()
    # an unconstrained context is unaffected
    ok = bbme.MF(f1, f2, search, block, 2)
    assert np.array_equal(ok.calcMotionBlockMatching(), expect)
    ok.close()


def test_kernel_limits_are_refused(bbme):
    """What the kernels do not take is refused when the context is created (BBME_ERR_UNSUPPORTED), not computed wrongly: a search
    range beyond 127 (the spiral ranks are 16-bit), a block size that is not a power of two in 2..64."""
    from blockbasedmotionestimation_amd import _capi
    f = np.zeros((256, 384), np.uint8)
    for search, block in (([16 + 2 * 128], [16]), ([40], [12]), ([20], [1]), ([200], [128])):
        with pytest.raises(_capi.BbmeError) as err:
            bbme.MF(f, f, search, block, 1)
        assert err.value.status == _capi.ERR_UNSUPPORTED, (search, block, err.value.message)
    # 2 x 2 blocks on a frame whose width is not a multiple of four: the kernels move rows as dwords
    with pytest.raises(_capi.BbmeError) as err:
        bbme.MF(f[:, :130], f[:, :130], [10], [2], 1)
    assert err.value.status == _capi.ERR_UNSUPPORTED and "multiples of 4" in err.value.message


def test_flat_and_zero_frames_tie_breaking(bbme, oracle):
    """All-equal SADs everywhere: the winner is decided purely by spiral order (search) and by
    candidate order (regulariser)."""
    z = np.zeros((128, 192), np.uint8)
    compare_stagewise(bbme, oracle, z, z, [48, 48], [16, 16])
    c = np.full((128, 192), 77, np.uint8)
    compare_stagewise(bbme, oracle, c, c, [30, 30], [16, 16])
    # flat frame1, textured frame2 and vice versa
    f1, f2, _ = bbme.synth_pair(192, 128, 5, max_motion=8)
    compare_stagewise(bbme, oracle, c, f2, [30, 30], [16, 16])
    compare_stagewise(bbme, oracle, f1, c, [30, 30], [16, 16])


def test_periodic_texture_many_ties(bbme, oracle):
    """Stripes and checkerboards: many candidates share the minimal SAD."""
    y, x = np.mgrid[0:192, 0:256]
    stripes = ((x // 4) % 2 * 200).astype(np.uint8)
    checker = (((x // 8) + (y // 8)) % 2 * 255).astype(np.uint8)
    compare_stagewise(bbme, oracle, stripes, np.roll(stripes, 3, axis=1), [48, 48], [16, 16])
    compare_stagewise(bbme, oracle, checker, np.roll(checker, (5, -2), axis=(0, 1)), [48, 48], [16, 16])


def test_large_motion_predictions_leave_image(bbme, oracle):
    """Coarse MVs doubled at the next level push predictions outside the image: those blocks
    take a zero MV without searching (motion_framework.cpp:304-310)."""
    rng = np.random.default_rng(7)
    f1 = rng.integers(0, 256, (256, 320), dtype=np.uint8)
    f2 = np.roll(f1, (37, -45), axis=(0, 1))
    compare_stagewise(bbme, oracle, f1, f2, [80, 80, 80], [16, 16, 16])


def test_stagewise_parity_with_a_one_wave_solver(bbme, oracle, monkeypatch):
    """Every intermediate MV grid when the solver behind the tile kernel is a single wave (the other extreme of the
    schedule: every cross-tile chain is walked sequentially), on content with many changes per sweep."""
    monkeypatch.setenv("BBME_SOLVE_WGS", "1")
    monkeypatch.setenv("BBME_SOLVE_WAVES", "1")
    f1, f2, _ = bbme.synth_pair(328, 200, 5151, max_motion=12)
    compare_stagewise(bbme, oracle, f1, f2, [48, 48, 48], [16, 16, 16])
    rng = np.random.default_rng(12)
    n1 = rng.integers(0, 256, (160, 224), dtype=np.uint8)
    n2 = rng.integers(0, 256, (160, 224), dtype=np.uint8)
    compare_stagewise(bbme, oracle, n1, n2, [40, 40], [8, 8])


def test_stagewise_parity_with_the_strip_form_of_pass_1(bbme, oracle, monkeypatch):
    """k_reg_pass1_strip (what batched contexts run pass 1 with on large grids of 2 x 2 and 4 x 4 blocks: four blocks per lane from a
    3 x 6 window, the blocks that need their images listed per wave and evaluated densely), forced onto a single-pair context and onto
    every grid: every intermediate MV grid against the oracle -- first sweeps (candidates from the parent grid) and second sweeps,
    grids whose width is a multiple of four blocks and grids that fall back to the plain form, content with many and with few
    non-uniform neighbourhoods."""
    monkeypatch.setenv("BBME_PASS1_STRIP", "1")
    monkeypatch.setenv("BBME_PASS1_LANES_MAX", "0")
    f1, f2, _ = bbme.synth_pair(384, 256, 6161, max_motion=14)
    compare_stagewise(bbme, oracle, f1, f2, [48, 48, 48], [16, 16, 16])
    f1, f2, _ = bbme.synth_pair(216, 136, 6162, max_motion=6)          # b = 4: 54 columns (plain form), b = 2: 108 (strips)
    compare_stagewise(bbme, oracle, f1, f2, [24, 24], [8, 8])
    rng = np.random.default_rng(13)
    n1 = rng.integers(0, 256, (128, 192), dtype=np.uint8)
    n2 = rng.integers(0, 256, (128, 192), dtype=np.uint8)
    compare_stagewise(bbme, oracle, n1, n2, [24, 24], [4, 4])         # unrelated frames: nearly every block takes the image path
    z = np.zeros((96, 160), np.uint8)
    compare_stagewise(bbme, oracle, z, z, [16, 16], [4, 4])           # flat: none does
    # with a relaxation launch behind pass 1 the strip kernel evaluates nothing itself (RegArgs::lazy): the blocks that need their
    # images keep their old value, are flagged, and the relaxation's first round evaluates them
    monkeypatch.setenv("BBME_RELAX_STEPS", "1")
    f1, f2, _ = bbme.synth_pair(384, 256, 6163, max_motion=14)
    compare_stagewise(bbme, oracle, f1, f2, [48, 48, 48], [16, 16, 16])
    compare_stagewise(bbme, oracle, n1, n2, [24, 24], [4, 4])
    monkeypatch.setenv("BBME_RELAX_STEPS", "3")
    compare_stagewise(bbme, oracle, n1, n2, [24, 24], [8, 8])


def test_stagewise_parity_with_relaxation_steps(bbme, oracle, monkeypatch):
    """Every intermediate MV grid with two relaxation launches forced into every sweep (k_reg_iter: the tile-resident
    local fixed point), on content with many changes per sweep and on every block size."""
    monkeypatch.setenv("BBME_RELAX_STEPS", "2")
    f1, f2, _ = bbme.synth_pair(328, 200, 5151, max_motion=12)
    compare_stagewise(bbme, oracle, f1, f2, [48, 48, 48], [16, 16, 16])
    rng = np.random.default_rng(12)
    n1 = rng.integers(0, 256, (160, 224), dtype=np.uint8)
    n2 = rng.integers(0, 256, (160, 224), dtype=np.uint8)
    compare_stagewise(bbme, oracle, n1, n2, [40, 40], [8, 8])
    compare_stagewise(bbme, oracle, n1, n2, [12, 12], [4, 4])
    f1, f2, _ = bbme.synth_pair(512, 512, 5152, max_motion=30)
    compare_stagewise(bbme, oracle, f1, f2, [64, 64, 64], [32, 32, 32])
    compare_stagewise(bbme, oracle, f1, f2, [80, 80], [64, 64])


def test_noise_frames(bbme, oracle):
    """Uncorrelated noise: regulariser candidates often point outside the image (FLT_MAX energy)."""
    rng = np.random.default_rng(11)
    f1 = rng.integers(0, 256, (192, 256), dtype=np.uint8)
    f2 = rng.integers(0, 256, (192, 256), dtype=np.uint8)
    compare_stagewise(bbme, oracle, f1, f2, [48, 48], [16, 16])
    compare_stagewise(bbme, oracle, f1, f2, [40, 40], [8, 8])


def test_full_pipeline_host_frames(bbme, oracle):
    """MF(image1, image2, ...).calcMotionBlockMatching() end to end, including the product's own
    host padding + pyrDown, against the oracle's."""
    f1, f2, _ = bbme.synth_pair(584, 388, 1010, max_motion=7)          # RubberWhale geometry, cfg1
    search, block = [30, 30, 30], [16, 16, 16]
    omf = oracle.OracleMF(f1, f2, search, block)
    exp = omf.calc_motion_block_matching()
    mf = bbme.MF(f1, f2, search, block, 3)
    assert (mf.padded_width, mf.padded_height, mf.padding_x, mf.padding_y) == (640, 448, 28, 30)
    for lvl in range(3):
        a, b = mf.get_level_planes(lvl)
        assert np.array_equal(a, omf.image(lvl, 1)) and np.array_equal(b, omf.image(lvl, 2))
    got = mf.calcMotionBlockMatching()
    assert got.dtype == np.float32 and got.shape == (448, 640, 2)
    assert np.array_equal(got, exp)
    # graph replay: a second call on the same context gives the same field
    assert np.array_equal(mf.calcMotionBlockMatching(), exp)
    cells = mf.get_cells()
    assert np.array_equal(cells.astype(np.float32), exp[::2, ::2])
    # caller-supplied output arrays (e.g. views of pinned memory)
    buf = np.zeros((448, 640, 2), np.float32)
    assert mf.get_flow(buf) is buf and np.array_equal(buf, exp)
    cbuf = np.zeros((224, 320, 2), np.int16)
    assert mf.get_cells(cbuf) is cbuf and np.array_equal(cbuf, cells)
    with pytest.raises(bbme.BbmeError):
        mf.get_flow(np.zeros((448, 640, 2), np.float64))
    with pytest.raises(bbme.BbmeError):
        mf.get_cells(np.zeros((224, 321, 2), np.int16))
    # the multi-GPU path: a cell grid copied elsewhere in HBM (as after a gather) expands to the same field
    import torch
    moved = torch.from_numpy(cells.copy()).cuda()
    dense = torch.zeros((448, 640, 2), dtype=torch.float32, device="cuda")
    mf.expand_cells_device(moved.data_ptr(), dense.data_ptr())
    mf.synchronize()
    assert np.array_equal(dense.cpu().numpy(), exp)
    assert np.array_equal(np.repeat(np.repeat(cells, 2, 0), 2, 1).astype(np.float32), exp)
    # ... and on a stream of the caller's (the one a gather completes on)
    side = torch.cuda.Stream()
    dense2 = torch.zeros((448, 640, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    mf.expand_cells_device(moved.data_ptr(), dense2.data_ptr(), side.cuda_stream)
    side.synchronize()
    assert np.array_equal(dense2.cpu().numpy(), exp)
    mf.close()


def test_device_frames_gpu_pyramid(bbme, oracle):
    """Frames already in HBM: zero padding and the pyrDown cascade as HIP kernels."""
    import torch
    f1, f2, _ = bbme.synth_pair(600, 410, 1020, max_motion=10)
    search, block = [48, 48, 48], [16, 16, 16]
    omf = oracle.OracleMF(f1, f2, search, block)
    t1 = torch.from_numpy(f1).cuda()
    t2 = torch.from_numpy(f2).cuda()
    mf = bbme.MF(t1, t2, search, block, 3, frames_on_device=True)
    mf.synchronize()
    for lvl in range(3):
        a, b = mf.get_level_planes(lvl)
        assert np.array_equal(a, omf.image(lvl, 1)), "image1 plane of level %d" % lvl
        assert np.array_equal(b, omf.image(lvl, 2)), "image2 plane of level %d" % lvl
    got = mf.calcMotionBlockMatching()
    assert np.array_equal(got, omf.calc_motion_block_matching())
    mf.close()


@pytest.mark.parametrize("w,h,search,block", [
    (250, 130, [30], [16]),                              # padded both ways, one level (border copy only)
    (100, 60, [12, 12, 12], [4, 4, 4]),                  # level widths 104 / 52 / 26: the one-pixel-per-thread pyrDown
    (1000, 700, [24, 24, 24, 24], [8, 8, 8, 8]),         # odd-looking size, four levels
    (1920, 1080, [48, 48, 48], [16, 16, 16]),            # cfg2
    (3840, 2160, [80, 80, 80, 80], [16, 16, 16, 16]),    # cfg3: the bench's geometry
])
def test_gpu_pyramid_planes_match_the_oracle(bbme, oracle, w, h, search, block):
    """MF::MF on the GPU (zero border :57-61, pyrDown cascade :86-106) for frames in HBM and for host frames (which take
    the same kernels): every plane of every level equals the oracle's, including rows of a strided (pitch > width) tensor."""
    import torch
    rng = np.random.default_rng(w * 7 + h)
    f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    f2 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    L = len(block)
    omf = oracle.OracleMF(f1, f2, search, block)
    wide = torch.zeros((2, h, w + 13), dtype=torch.uint8, device="cuda")
    wide[0, :, :w] = torch.from_numpy(f1).cuda()
    wide[1, :, :w] = torch.from_numpy(f2).cuda()
    for frames_on_device in (True, False):
        if frames_on_device:
            mf = bbme.MF(wide[0, :, :w], wide[1, :, :w], search, block, L, frames_on_device=True)
        else:
            mf = bbme.MF(f1, f2, search, block, L)
        for lvl in range(L):
            a, b = mf.get_level_planes(lvl)
            assert np.array_equal(a, omf.image(lvl, 1)), "image1 plane of level %d (device frames: %s)" % (lvl, frames_on_device)
            assert np.array_equal(b, omf.image(lvl, 2)), "image2 plane of level %d (device frames: %s)" % (lvl, frames_on_device)
        mf.close()
    omf.close()


def _write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5\n# bbme test frame\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(np.ascontiguousarray(img, np.uint8).tobytes())


def test_cli_reproduces_reference_driver_sequence(bbme, oracle, tmp_path):
    """bbme_cli (C++ host side: MF / Flow classes over the C-ABI) runs main_class.cpp's sequence:
    4x bilinear up-sampling, MF, calcMotionBlockMatching, strip padding + every 4th pixel / 4,
    MotionToColor + image file, WriteFlowFile, CalculateMSE.  Checked against the same sequence on the oracle."""
    import subprocess
    from blockbasedmotionestimation_amd import build as _build
    f1, f2, _ = bbme.synth_pair(146, 97, 77, max_motion=3)
    _write_pgm(str(tmp_path / "a.pgm"), f1)
    _write_pgm(str(tmp_path / "b.pgm"), f2)
    gt = np.zeros((97, 146, 2), np.float32)
    gt[..., 0] = 0.75
    gt[5:9, 5:9] = 1.666666752e9
    bbme.Flow().WriteFlowFile(gt, str(tmp_path / "gt.flo"))
    out = tmp_path / "out.flo"
    r = subprocess.run([_build.CLI, str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), "--levels", "3", "--block", "16",
                        "--search", "30", "--out", str(out), "--gt", str(tmp_path / "gt.flo"),
                        "--color", str(tmp_path / "flow.ppm")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    u1, u2 = oracle.resize_linear_x4(f1), oracle.resize_linear_x4(f2)
    omf = oracle.OracleMF(u1, u2, [30] * 3, [16] * 3)
    flow = omf.calc_motion_block_matching()
    exp = oracle.subsample_div4(flow, omf.padding_x, omf.padding_y, 146, 97)
    got = bbme.Flow().ReadFlowFile(str(out))
    assert np.array_equal(got, exp)
    mse = float(r.stdout.split("Calculated MSE is")[1].split()[0])
    assert mse == pytest.approx(oracle.calculate_mse(gt, exp), rel=1e-8)
    img, rng = oracle.motion_to_color(exp)                      # main_class.cpp:73-75
    ppm = (tmp_path / "flow.ppm").read_bytes()
    head = b"P6\n146 97\n255\n"
    assert ppm.startswith(head)
    assert np.array_equal(np.frombuffer(ppm[len(head):], np.uint8).reshape(97, 146, 3), img[..., ::-1])
    assert ("max motion: %.4f  motion range: u = %.3f .. %.3f;  v = %.3f .. %.3f" % rng) in r.stdout


def test_epe_against_middlebury_ground_truth_warped_pair(bbme, oracle):
    """BASELINE configs[0] asks for EPE against Middlebury ground truth.  The Middlebury frames are
    not in the reference (only the GT .flo files are), so the pair is synthetic: frame2 is a texture
    and frame1(x) = frame2(x + gt(x)) by bilinear sampling, which makes the Venus GT file the true
    flow of the pair.  The reference's own pipeline and literals (4x up-sampling, 4 levels, block 32,
    search 64) must then give the same EPE on the HIP path and on the oracle, and a small one."""
    import os
    from conftest import GOLDEN
    gt = bbme.Flow().ReadFlowFile(os.path.join(GOLDEN, "gt_Venus_flow10.flo"))
    h, w = gt.shape[:2]
    frame1, frame2 = bbme.warp_pair_from_flow(gt)
    search, block = [64] * 4, [32] * 4
    u1, u2 = bbme.resize_x4(frame1), bbme.resize_x4(frame2)
    assert np.array_equal(u1, oracle.resize_linear_x4(frame1))
    mf = bbme.MF(u1, u2, search, block, 4)
    flow = mf.calcMotionBlockMatching()
    sub = bbme.subsample_div4(flow, mf.padding_x, mf.padding_y, w, h)
    epe = bbme.Flow().CalculateMSE(gt, sub)
    omf = oracle.OracleMF(u1, u2, search, block)
    osub = oracle.subsample_div4(omf.calc_motion_block_matching(), omf.padding_x, omf.padding_y, w, h)
    assert np.array_equal(sub, osub)
    assert epe == oracle.calculate_mse(gt, osub)
    assert epe < 0.6, "average end-point error %.3f px" % epe
    # the same number without downloading the field: EPE reduced on the device from the 2x2-cell grid.  Per-pixel
    # arithmetic is the reference's; only the order of the double sum differs -> 1e-12 relative
    import torch
    gt_holes = gt.copy()
    gt_holes[7:19, 30:60] = 1.666666752e9
    gt_holes[100, 100, 0] = np.nan
    for g in (gt, gt_holes):
        dev = mf.calculate_mse_device(torch.from_numpy(g).cuda(), scale=4)
        assert dev == pytest.approx(oracle.calculate_mse(g, osub), rel=1e-12)
    with pytest.raises(bbme.BbmeError):
        mf.calculate_mse_device(torch.zeros((h + 8, w, 2), dtype=torch.float32, device="cuda"), scale=4)
    mf.close()


def test_errors_through_the_boundary(bbme):
    z = np.zeros((64, 64), np.uint8)
    with pytest.raises(bbme.BbmeError) as e:          # one block per dimension at the coarsest level
        bbme.MF(z, z, [30, 30], [16, 32])
    assert e.value.status == -4
    with pytest.raises(bbme.BbmeError) as e:          # no multiple below twice the size
        bbme.MF(np.zeros((40, 40), np.uint8), np.zeros((40, 40), np.uint8), [80, 80, 80], [32, 32, 32])
    assert e.value.status == -2
    with pytest.raises(bbme.BbmeError):
        bbme.MF(z, np.zeros((64, 32), np.uint8), [30], [16])
    mf = bbme.MF(np.zeros((128, 128), np.uint8), np.zeros((128, 128), np.uint8), [30, 30], [16, 16])
    with pytest.raises(bbme.BbmeError) as e:          # level 0 searched before level 1 is done
        mf.stage_search(0)
    assert e.value.status == -7
    mf.stage_search(1)
    with pytest.raises(bbme.BbmeError) as e:          # sweep at 4 while the grid is at 16
        mf.stage_regularize(1, 4, 1)
    assert e.value.status == -7
    mf.close()


def test_1080p_pair_against_oracle(bbme, oracle):
    """BASELINE configs[1] at full size (1920x1080, 16x16, +-16, 3 levels) against the oracle, a few seconds of CPU:
    130 560 / 522 240 blocks at b = 4 / 2: hundreds of tiles per sweep, work lists of thousands of blocks."""
    f1, f2, _ = bbme.synth_pair(1920, 1080, 1020, max_motion=12)
    search, block = [48] * 3, [16] * 3
    omf = oracle.OracleMF(f1, f2, search, block, use_cache=False)
    exp = omf.calc_motion_block_matching()
    mf = bbme.MF(f1, f2, search, block, 3)
    got = mf.calcMotionBlockMatching()
    again = mf.calcMotionBlockMatching()
    mf.close()
    omf.close()
    assert np.array_equal(got, exp) and np.array_equal(again, exp)


@pytest.mark.parametrize("search_size,block_size,seed", [(80, 16, 5), (72, 8, 6)], ids=["cfg3_b16", "cfg4_b8"])
def test_4k_uncorrelated_frames_against_oracle(bbme, oracle, search_size, block_size, seed):
    """BASELINE configs[2] and configs[3] at full size (4K, 4 levels, +-32; 16x16 blocks, and 8x8 blocks = 130 560
    level-0 macroblocks, motion_framework.cpp:226-244) on the worst content for the regulariser: two unrelated noise
    frames, so that nearly every block changes in nearly every sweep (long queues, full work lists, overflow list).
    The oracle needs 5-15 seconds here because its searches hit the image border early."""
    rng = np.random.default_rng(seed)
    f1 = rng.integers(0, 256, (2160, 3840), dtype=np.uint8)
    f2 = rng.integers(0, 256, (2160, 3840), dtype=np.uint8)
    search, block = [search_size] * 4, [block_size] * 4
    mf = bbme.MF(f1, f2, search, block, 4)
    got = mf.calcMotionBlockMatching()
    flag, _ = mf.last_sweep_passes()
    mf.close()
    omf = oracle.OracleMF(f1, f2, search, block, use_cache=False)
    exp = omf.calc_motion_block_matching()
    omf.close()
    assert flag == 0, "a sweep hit its round cap"
    assert np.array_equal(got, exp)


def test_pipelined_sequence_matches_oracle_per_pair(bbme, oracle):
    """A sequence on one GPU with several pairs in flight (contexts re-used round-robin with new frames):
    every pair's field is the oracle's, whatever else is running beside it."""
    from blockbasedmotionestimation_amd.sequence import estimate_pairs_pipelined
    search, block = [40, 40, 40], [8, 8, 8]
    pairs = [bbme.synth_pair(328, 200, 4200 + i, max_motion=10)[:2] for i in range(7)]
    expect = []
    for f1, f2 in pairs:
        omf = oracle.OracleMF(f1, f2, search, block)
        full = omf.calc_motion_block_matching()
        py, px = omf.padding_y, omf.padding_x
        expect.append(full[py:py + 200, px:px + 328])
        omf.close()
    for k, per in ((1, 1), (3, 1), (4, 2), (6, 3), (16, 2), (16, 8)):      # pairs in flight, pairs per batched context
        got = estimate_pairs_pipelined(pairs, search, block, in_flight=k, batch=per)
        assert len(got) == len(pairs)
        for g, e in zip(got, expect):
            assert g.shape == (200, 328, 2) and np.array_equal(g, e)
    assert estimate_pairs_pipelined([], search, block) == []
    mf = bbme.MF(pairs[0][0], pairs[0][1], search, block, 3)
    with pytest.raises(bbme.BbmeError):
        mf.set_frames(pairs[0][0][:100], pairs[0][1][:100])
    mf.close()


@pytest.mark.parametrize("w,h,search,block,levels", [(328, 200, 40, 8, 3), (400, 304, 48, 16, 2), (712, 488, 80, 16, 3)])
def test_batched_context_matches_the_oracle_pair_by_pair(bbme, oracle, w, h, search, block, levels):
    """bbme_create_batch: several pairs behind one launch sequence (every kernel gets the pair as a second grid dimension,
    every per-pair buffer a pair stride).  Each pair's field must be the oracle's -- the pairs share nothing
    (motion_framework.h:37-46) -- with the speculative search on (the default) and off, after replacing one pair's frames,
    and the compact cells must agree with the dense fields."""
    ss, bs = [search] * levels, [block] * levels
    pairs = [bbme.synth_pair(w, h, 5100 + i, max_motion=9 + 2 * i)[:2] for i in range(5)]

    def expect(f1, f2):
        omf = oracle.OracleMF(f1, f2, ss, bs)
        out = omf.calc_motion_block_matching()
        omf.close()
        return out
    exp = [expect(f1, f2) for f1, f2 in pairs]
    mb = bbme.MFBatch(pairs[:4], ss, bs, levels)
    assert mb.batch == 4
    for spec in (True, False):
        mb.set_speculation(spec)
        got = mb.calcMotionBlockMatching()
        for p in range(4):
            assert np.array_equal(got[p], exp[p]), "pair %d (speculation %s)" % (p, spec)
            cells = mb.get_pair_cells(p)
            assert np.array_equal(np.repeat(np.repeat(cells, 2, 0), 2, 1).astype(np.float32), got[p])
    mb.set_pair(2, *pairs[4])                                   # a new pair into slot 2; the others keep their frames
    got = mb.calcMotionBlockMatching()
    for p, e in enumerate([exp[0], exp[1], exp[4], exp[3]]):
        assert np.array_equal(got[p], e), "pair %d after replacing pair 2" % p
    with pytest.raises(bbme.BbmeError):
        mb.get_pair_flow(4)
    mb.close()
    # frames already in HBM (torch tensors), as bench.py's sequence leg hands them over
    import torch
    dev = [(torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()) for f1, f2 in pairs[:3]]
    mb = bbme.MFBatch(dev, ss, bs, levels, frames_on_device=True)
    got = mb.calcMotionBlockMatching()
    mb.close()
    for p in range(3):
        assert np.array_equal(got[p], exp[p]), "pair %d (device frames)" % p


@pytest.mark.parametrize("variant", ["generic_search", "raster", "block2", "block4", "block64"])
def test_batched_context_with_the_generic_search_kernels(bbme, oracle, monkeypatch, variant):
    """The pair as blockIdx.y of k_search_generic (block 2, 4 and 64, the raster find_min_block variant, and BBME_GENERIC_SEARCH=1
    on block sizes that normally take the strip kernel): every pair of a batched context against the oracle."""
    if variant == "block4":
        w, h, ss, bs = 200, 136, [12, 12], [4, 4]
    elif variant == "block2":
        w, h, ss, bs = 200, 136, [10, 12], [2, 4]
    elif variant == "block64":
        w, h, ss, bs = 512, 384, [80, 80], [64, 64]
    else:
        w, h, ss, bs = 296, 200, [30, 40], [16, 8]
    if variant == "generic_search":
        monkeypatch.setenv("BBME_GENERIC_SEARCH", "1")
    pairs = [bbme.synth_pair(w, h, 5300 + i, max_motion=5 + i)[:2] for i in range(3)]
    mb = bbme.MFBatch(pairs, ss, bs, len(bs))
    monkeypatch.delenv("BBME_GENERIC_SEARCH", raising=False)
    mb.set_search_mode(variant == "raster")
    got = mb.calcMotionBlockMatching()
    mb.close()
    for p, (f1, f2) in enumerate(pairs):
        omf = oracle.OracleMF(f1, f2, ss, bs)
        omf.set_raster_search(variant == "raster")
        exp = omf.calc_motion_block_matching()
        omf.close()
        assert np.array_equal(got[p], exp), "pair %d (%s)" % (p, variant)


def test_batched_context_refuses_single_pair_calls_and_wrong_tensors(bbme):
    """The stage calls, the plane injection, the sweep counters and the device-side EPE address one pair: a batched context
    refuses them (BBME_ERR_UNSUPPORTED) instead of working on pair 0.  A device tensor of another shape than the context's
    frames is refused before its pointer reaches the padding kernel, and the inherited setters address pair 0."""
    import torch
    from blockbasedmotionestimation_amd import _capi
    w, h = 200, 136
    pairs = [bbme.synth_pair(w, h, 5400 + i, max_motion=4)[:2] for i in range(2)]
    mb = bbme.MFBatch(pairs, [30, 30], [16, 16], 2)
    for call in (lambda: mb.stage_search(1), lambda: mb.stage_regularize(1, 16, 1), lambda: mb.stage_get_mvs(1, 16),
                 lambda: mb.stage_expand(), lambda: mb.sweep_stats(), lambda: mb.last_sweep_passes(),
                 lambda: mb.set_level_planes(0, *mb.get_level_planes(0)),
                 lambda: mb.calculate_mse_device(torch.zeros((h, w, 2), dtype=torch.float32, device="cuda"), 1)):
        with pytest.raises(_capi.BbmeError) as err:
            call()
        assert err.value.status == _capi.ERR_UNSUPPORTED, err.value.message
    good = torch.from_numpy(pairs[0][0]).cuda()
    for bad in (torch.zeros((h - 8, w), dtype=torch.uint8, device="cuda"), torch.zeros((h, w), dtype=torch.int8, device="cuda"),
                torch.zeros((1, h, w), dtype=torch.uint8, device="cuda")):
        with pytest.raises(_capi.BbmeError) as err:
            mb.set_pair_device(1, good, bad)
        assert err.value.status == _capi.ERR_INVALID
    with pytest.raises(_capi.BbmeError):
        mb.set_pair_device(2, good, good)
    ref = mb.calcMotionBlockMatching()
    # the inherited entry points without a pair index address pair 0 and keep the bookkeeping intact
    mb.set_frames_device(torch.from_numpy(pairs[1][0]).cuda(), torch.from_numpy(pairs[1][1]).cuda())
    mb.set_pair_device(1, torch.from_numpy(pairs[0][0]).cuda(), torch.from_numpy(pairs[0][1]).cuda())
    swapped = mb.calcMotionBlockMatching()
    assert np.array_equal(swapped[0], ref[1]) and np.array_equal(swapped[1], ref[0])
    mb.set_frames(*pairs[0])
    assert np.array_equal(mb.calcMotionBlockMatching()[0], ref[0])
    mb.close()


def test_deep_batches_match_the_oracle(bbme, oracle):
    """The largest batch a context takes (BBME_MAX_BATCH = 64 pairs behind every launch; bench.py's `sequence_deep` legs) and
    two deep contexts side by side on their own streams: every pair's field must be the oracle's.  Twelve distinct pairs and
    the same pairs rolled by a few pixels (new content for the GPU, one more oracle run each only for the ones checked)."""
    w, h, levels = 296, 200, 3
    ss, bs = [30] * levels, [16] * levels
    base = [bbme.synth_pair(w, h, 5200 + i, max_motion=6 + i)[:2] for i in range(12)]
    pairs = []
    for k in range(64):
        f1, f2 = base[k % 12]
        sh = (3 * (k // 12), 5 * (k // 12))
        pairs.append((np.ascontiguousarray(np.roll(f1, sh, (0, 1))), np.ascontiguousarray(np.roll(f2, sh, (0, 1)))))

    def expect(k):
        omf = oracle.OracleMF(pairs[k][0], pairs[k][1], ss, bs)
        out = omf.calc_motion_block_matching()
        omf.close()
        return out
    checked = list(range(12)) + [12, 25, 38, 51, 63]
    exp = {k: expect(k) for k in checked}
    mb = bbme.MFBatch(pairs, ss, bs, levels)
    assert mb.batch == 64
    got = mb.calcMotionBlockMatching()
    for k in checked:
        assert np.array_equal(got[k], exp[k]), "pair %d of 64" % k
    mb.close()
    with pytest.raises(bbme.BbmeError):
        bbme.MFBatch(pairs + pairs[:1], ss, bs, levels)            # 65 pairs: more than BBME_MAX_BATCH
    a, b = bbme.MFBatch(pairs[:32], ss, bs, levels), bbme.MFBatch(pairs[32:], ss, bs, levels)
    for _ in range(3):
        a.estimate_async()
        b.estimate_async()
    ga, gb = a.calcMotionBlockMatching(), b.calcMotionBlockMatching()
    for k in checked:
        g = ga[k] if k < 32 else gb[k - 32]
        assert np.array_equal(g, exp[k]), "pair %d (two contexts of 32)" % k
    a.close()
    b.close()


@pytest.mark.parametrize("cfg", ["cfg2_1080p", "cfg3_4k", "cfg4_4k_b8"])
def test_full_size_properties(bbme, cfg):
    """BASELINE.json's full sizes, through properties that need no oracle run:
    (1) determinism / replay, (2) the dense field is constant on 2x2 cells and equals the compact
    cells, (3) a globally translated frame yields exactly that translation wherever the block and
    its whole neighbourhood stay inside both frames, (4) every MV is an integer."""
    w, h, search, block = {
        "cfg2_1080p": (1920, 1080, [48] * 3, [16] * 3),
        "cfg3_4k": (3840, 2160, [80] * 4, [16] * 4),
        "cfg4_4k_b8": (3840, 2160, [72] * 4, [8] * 4),
    }[cfg]
    rng = np.random.default_rng(99)
    big = rng.integers(0, 256, (h + 64, w + 64), dtype=np.uint8)
    dx, dy = 5, -3
    f1 = big[32:32 + h, 32:32 + w]
    f2 = big[32 - dy:32 - dy + h, 32 - dx:32 - dx + w]       # f2(y + dy, x + dx) == f1(y, x)
    mf = bbme.MF(f1, f2, search, block, len(block))
    a = mf.calcMotionBlockMatching()
    b = mf.calcMotionBlockMatching()
    assert np.array_equal(a, b)
    assert np.array_equal(a, np.round(a))
    cells = mf.get_cells()
    assert np.array_equal(np.repeat(np.repeat(cells, 2, 0), 2, 1).astype(np.float32), a)
    py, px = mf.padding_y, mf.padding_x
    inner = a[py + 128:py + h - 128, px + 128:px + w - 128]
    frac = np.mean((inner[..., 0] == dx) & (inner[..., 1] == dy))
    assert frac == 1.0, "only %.4f of interior pixels carry the true translation" % frac
    mf.close()


@pytest.mark.parametrize("overlap,speculate", [(True, False), (False, True), (True, True)],
                         ids=["second_stream", "in_order_speculative", "second_stream_speculative"])
def test_cell_gather_world_size_one_nccl(bbme, oracle, overlap, speculate):
    """The multi-GPU step (sequence.CellGather / mf_cell_gather, what `bench.py --gpus N` runs) on the one GPU of this
    box with an RCCL process group of size one: estimate on the work stream, then either the cell grid staged, gathered and
    expanded with the expand kernel on the side stream (three steps: both staging buffers, overlap of gather and next
    estimate) or gather and expansion in order on the work stream; rank 0's dense field must be the oracle's."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from blockbasedmotionestimation_amd.sequence import mf_cell_gather
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    f1, f2, _ = bbme.synth_pair(640, 360, 6100, max_motion=14)
    search, block = [48, 48, 48], [16, 16, 16]
    omf = oracle.OracleMF(f1, f2, search, block)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        mf = bbme.MF(f1, f2, search, block, 3)
        mf.set_speculation(speculate)
        for lvl in range(3):
            mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
        exp = omf.calc_motion_block_matching()
        with torch.cuda.stream(torch.cuda.Stream(device=0)):
            g = mf_cell_gather(mf, 0, overlap=overlap)
            for _ in range(3):
                g.step()
            g.fence()
            got = g.flows[0].cpu().numpy()
        assert g.world == 1 and got.shape == exp.shape
        assert np.array_equal(got, exp)
        assert np.array_equal(mf.get_flow(), exp)
        mf.close()
    finally:
        dist.destroy_process_group()
        omf.close()


RASTER_CASES = [
    (320, 208, [30, 30, 30], [16, 16, 16], 2001, 12),           # B=16, R=7, 3 levels
    (384, 256, [48, 48], [16, 16], 2002, 24),                   # R=16
    (256, 256, [72, 72], [8, 8], 2003, 20),                     # B=8, R=32: windows and predictions leave the image
    (512, 512, [64, 64, 64], [32, 32, 32], 2004, 30),           # the reference's literals
    (256, 192, [12, 12], [4, 4], 2005, 5),                      # B=4
    (256, 128, [17, 21], [16, 16], 2006, 3),                    # odd search - block
]


@pytest.mark.parametrize("w,h,search,block,seed,mm", RASTER_CASES)
def test_raster_search_variant_stagewise(bbme, oracle, w, h, search, block, seed, mm):
    """MF::find_min_block (motion_framework.cpp:246-294, the raster search whose call is commented out at
    :235) as the level's search -- clamped window, ties to the candidate nearest the block's own position, then raster
    order -- every stage against the oracle's restatement of it."""
    f1, f2, _ = bbme.synth_pair(w, h, seed, max_motion=mm)
    compare_stagewise(bbme, oracle, f1, f2, search, block, raster=True)


def test_raster_search_ties_and_outside_predictions(bbme, oracle):
    """Raster mode where its rules differ from the spiral's: flat frames (every SAD equal: the winner is the candidate
    nearest the zero-MV position), periodic texture, and coarse motion that throws predictions outside the image (the
    window is clamped; an empty window leaves the prediction as the result)."""
    z = np.full((128, 192), 90, np.uint8)
    compare_stagewise(bbme, oracle, z, z, [48, 48], [16, 16], raster=True)
    y, x = np.mgrid[0:192, 0:256]
    stripes = ((x // 4) % 2 * 200).astype(np.uint8)
    compare_stagewise(bbme, oracle, stripes, np.roll(stripes, 3, axis=1), [48, 48], [16, 16], raster=True)
    rng = np.random.default_rng(17)
    f1 = rng.integers(0, 256, (256, 320), dtype=np.uint8)
    f2 = np.roll(f1, (37, -45), axis=(0, 1))
    compare_stagewise(bbme, oracle, f1, f2, [80, 80, 80], [16, 16, 16], raster=True)
    # full pipeline through the graph (speculative search + fix-up) in raster mode
    omf = oracle.OracleMF(f1, f2, [80, 80, 80], [16, 16, 16])
    omf.set_raster_search(True)
    exp = omf.calc_motion_block_matching()
    mf = bbme.MF(f1, f2, [80, 80, 80], [16, 16, 16], 3)
    mf.set_search_mode(True)
    for lvl in range(3):
        mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
    assert np.array_equal(mf.calcMotionBlockMatching(), exp)
    mf.set_search_mode(False)                                   # and back: the spiral result differs and is the plain oracle's
    omf.close()
    omf = oracle.OracleMF(f1, f2, [80, 80, 80], [16, 16, 16])
    assert np.array_equal(mf.calcMotionBlockMatching(), omf.calc_motion_block_matching())
    mf.close(); omf.close()


def test_cpp_sequence_driver_over_rccl(bbme, oracle, tmp_path):
    """bbme_seq: the multi-GPU sequence without torch -- contexts per GPU, asynchronous uploads from pinned frames,
    ncclGather of the cell grids through libbbme_rccl.so (bbme_gather_cells) into double-buffered receive buffers, download on
    a copy stream beside the next round, asynchronous .flo writer (csrc/seq_schedule.hpp) -- here with the one GPU of this box
    (an RCCL communicator of size one, five pairs = five rounds, so that both buffers are re-used twice): every file equals
    the oracle's field, and the driver prints its per-round phase times."""
    import subprocess
    from blockbasedmotionestimation_amd import build as _build
    search, block, n = 40, 8, 5
    files, expect = [], []
    for p in range(n):
        f1, f2, _ = bbme.synth_pair(328, 200, 8800 + p, max_motion=10)
        for tag, img in (("a", f1), ("b", f2)):
            path = str(tmp_path / ("%d%s.pgm" % (p, tag)))
            _write_pgm(path, img)
            files.append(path)
        omf = oracle.OracleMF(f1, f2, [search] * 3, [block] * 3)
        full = omf.calc_motion_block_matching()
        expect.append(full[omf.padding_y:omf.padding_y + 200, omf.padding_x:omf.padding_x + 328])
        omf.close()
    r = subprocess.run([_build.SEQ, "--gpus", "1", "--levels", "3", "--block", str(block), "--search", str(search),
                        "--out", str(tmp_path)] + files, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "%d pairs of 328x200 on 1 GPU(s)" % n in r.stdout
    assert r.stdout.count("upload+pyramid") == n and "round %d:" % (n - 1) in r.stdout      # per-phase times of every round
    for p in range(n):
        got = bbme.Flow().ReadFlowFile(str(tmp_path / ("%04d.flo" % p)))
        assert np.array_equal(got, expect[p]), "pair %d" % p


def test_expand_gathered_grid(bbme):
    """bbme_expand_gathered (include/bbme_rccl.h): grid `rank` of a gather buffer -> the dense padded field, on the
    context's stream -- here on a hand-made two-rank buffer whose second grid is this context's own result."""
    import ctypes as C
    import torch
    from blockbasedmotionestimation_amd import build as _build
    rccl = C.CDLL(_build.RCCL_LIB)
    rccl.bbme_expand_gathered.restype = C.c_int
    rccl.bbme_expand_gathered.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    f1, f2, _ = bbme.synth_pair(200, 136, 61, max_motion=6)
    mf = bbme.MF(f1, f2, [24, 24], [8, 8], 2)
    want = mf.calcMotionBlockMatching().copy()
    cells = torch.from_numpy(mf.get_cells().copy()).cuda()
    words = cells.view(torch.int32).reshape(cells.shape[0], cells.shape[1])
    recv = torch.stack([torch.zeros_like(words), words]).contiguous()
    flow = torch.empty((mf.padded_height, mf.padded_width, 2), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for rank, expect in ((1, want), (0, np.zeros_like(want))):
        assert rccl.bbme_expand_gathered(mf._ctx, recv.data_ptr(), rank, flow.data_ptr()) == 0
        mf.synchronize()
        assert np.array_equal(flow.cpu().numpy(), expect)
    assert rccl.bbme_expand_gathered(mf._ctx, None, 0, flow.data_ptr()) != 0
    mf.close()


def test_jacobi_fast_mode_is_what_it_says(bbme, oracle):
    """the opt-in, NOT bit-exact regulariser mode.  Its definition -- every block of a sweep evaluated
    against the field as the previous sweep left it -- is restated in the oracle (jacobi_regularizer), and the kernels
    must produce exactly that field; it must differ from the reference's field on content where sweeps propagate, and
    switching back must give the reference's field again."""
    f1, f2, _ = bbme.synth_pair(512, 384, 3100, max_motion=20)
    search, block = [48, 48, 48], [16, 16, 16]
    omf = oracle.OracleMF(f1, f2, search, block)
    exact = omf.calc_motion_block_matching().copy()
    omf.close()
    omf = oracle.OracleMF(f1, f2, search, block)
    omf.set_jacobi_regularizer(True)
    jac = omf.calc_motion_block_matching().copy()
    mf = bbme.MF(f1, f2, search, block, 3)
    for lvl in range(3):
        mf.set_level_planes(lvl, omf.image(lvl, 1), omf.image(lvl, 2))
    omf.close()
    mf.set_regularizer_mode(True)
    got = mf.calcMotionBlockMatching()
    assert np.array_equal(got, jac)
    assert not np.array_equal(jac, exact)
    mf.set_regularizer_mode(False)
    assert np.array_equal(mf.calcMotionBlockMatching(), exact)
    mf.close()


def test_8k_pair_schedule_independence(bbme):
    """7680x4320, five levels (8.4 M 2x2 cells at level 0: every buffer, grid and list at four times the bench's size).  Too
    large for the oracle to check in seconds, so the properties that hold for any correct schedule: the same field on a
    second run, with the speculative search off, with the relaxation launches off, and with a single-wave solver; and the Jacobi mode differs."""
    import os
    f1, f2, _ = bbme.synth_pair(7680, 4320, 77, max_motion=24)
    search, block = [80] * 5, [16] * 5
    mf = bbme.MF(f1, f2, search, block, 5)
    a = mf.calcMotionBlockMatching()
    assert np.array_equal(mf.calcMotionBlockMatching(), a)
    mf.set_speculation(False)
    assert np.array_equal(mf.calcMotionBlockMatching(), a)
    mf.set_relaxation(False)
    assert np.array_equal(mf.calcMotionBlockMatching(), a)
    mf.set_relaxation(True)
    mf.set_regularizer_mode(True)
    assert not np.array_equal(mf.calcMotionBlockMatching(), a)
    mf.close()
    assert (a == np.round(a)).all() and np.array_equal(a[::2, ::2], a[1::2, 1::2])       # integer MVs, constant on 2x2 cells
    os.environ["BBME_SOLVE_WGS"], os.environ["BBME_SOLVE_WAVES"] = "8", "1"
    try:
        mf = bbme.MF(f1, f2, search, block, 5)
    finally:
        del os.environ["BBME_SOLVE_WGS"], os.environ["BBME_SOLVE_WAVES"]
    assert np.array_equal(mf.calcMotionBlockMatching(), a)
    mf.close()


def _flood_field(rows, cols, u, v, seeds, obstacles, rng):
    """An MV grid for the memo test: value u everywhere, the frame's true motion v at a few seed blocks (it floods right and down
    from each of them during a sweep), and obstacle blocks with other vectors (non-uniform neighbourhoods: their SADs are
    memoised by pass 1, the uniform blocks' are not)."""
    g = np.empty((rows, cols, 2), np.int16)
    g[...] = u
    for r, c in seeds:
        g[r, c] = v
    for _ in range(obstacles):
        g[int(rng.integers(0, rows)), int(rng.integers(0, cols))] = (int(rng.integers(-3, 4)), int(rng.integers(-3, 4)))
    return g


@pytest.mark.parametrize("b", [8, 16, 32])
def test_sad_memo_hits_and_misses(bbme, oracle, monkeypatch, b):
    """The SAD memo of the regulariser's chain form (b >= 8) on content made for it: frame2 is frame1 moved by v, the field
    holds u != v except at a few seeds, so v floods over the grid in ONE sweep -- every block behind the front is
    re-evaluated as its L, UL, U, UR inputs change one, two or four at a time, the first time with SADs nobody has summed yet
    (misses, summed by the lane group), later with SADs a neighbour's change forwarded or an earlier evaluation left (hits).
    Both sweeps at the block size, against the oracle's raster sweep, with the memo off, on, on with forwarding, with a
    one-wave solver (one wave re-uses its own slots round after round) and with wide rounds mixed in; the counters must show
    that hits and misses both happened."""
    rng = np.random.default_rng(900 + b)
    rows, cols = 12, 18
    h, w = rows * b, cols * b
    f1 = rng.integers(0, 256, (h, w), dtype=np.uint8)
    v, u = (2, -1), (0, 0)
    f2 = np.roll(f1, (v[1], v[0]), axis=(0, 1))                    # block at (x, y) of frame1 lies at (x + 2, y - 1) of frame2
    search, block = [b + 8], [b]
    fields = [_flood_field(rows, cols, u, v, [(1, 1)], 0, rng),                       # one flood, uniform territory
              _flood_field(rows, cols, u, v, [(0, 0), (3, 9), (7, 2)], 6, rng),       # three fronts that meet, obstacles
              _flood_field(rows, cols, v, u, [(2, 3)], 10, rng),                      # nothing to flood: obstacles only
              rng.integers(-2, 3, (rows, cols, 2)).astype(np.int16)]                  # every neighbourhood different
    lam = float(b // 2)
    # (by default the memo serves b >= 16 and forwards nothing -- what measured fastest; the test takes it down to b = 8 and
    # runs with and without forwarding)
    for env in ({}, {"BBME_MEMO": "0"}, {"BBME_MEMO_FORWARD": "1"}, {"BBME_SOLVE_WGS": "1", "BBME_SOLVE_WAVES": "1"},
                {"BBME_WIDE_THRESHOLD": "4", "BBME_MEMO_FORWARD": "1"}):
        env = dict({"BBME_MEMO": "1", "BBME_MEMO_FORWARD": "0"}, **env, BBME_MEMO_MIN_B="8")   # whatever the suite runs under
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        mf = bbme.MF(f1, f2, search, block, 1)
        for k in env:
            monkeypatch.delenv(k)
        omf = oracle.OracleMF(f1, f2, search, block)
        mf.set_level_planes(0, omf.image(0, 1), omf.image(0, 2))
        lookups = misses = 0
        for field in fields:
            mf.stage_set_mvs(0, b, field)
            omf.flow(0)[...] = 0
            omf.flow(0)[::b, ::b, :] = field
            omf.set_block_size(0, b)
            omf.set_lambda(0, lam)
            for mult in (1, 2):
                mf.stage_regularize(0, b, mult)
                omf.regularize_mvs(0, mult)
                got, exp = mf.stage_get_mvs(0, b).astype(np.int32), omf.block_mvs(0, b)
                assert np.array_equal(got, exp), "b=%d env=%s sweep %d: %d blocks differ" % (b, env, mult, int((got != exp).any(-1).sum()))
                st = mf.sweep_stats()
                lookups += st[9]; misses += st[10]
        # the flood really happened (the test's premise), and the memo was exercised both ways
        if env.get("BBME_MEMO") == "0":
            assert lookups == 0
        else:
            assert lookups > 0 and 0 < misses < lookups, (b, env, lookups, misses)
        mf.close()
        omf.close()
