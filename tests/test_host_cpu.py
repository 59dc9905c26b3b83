"""CPU tests of the product's host side (no GPU): the C-ABI library loads and exports every symbol
include/bbme.h declares; MF::MF's padding / pyramid, the .flo codec and the EPE in libbbme.so agree
with the oracle and with the reference-made fixtures; error behaviour without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol(bbme):
    from blockbasedmotionestimation_amd import _capi
    header = open(os.path.join(ROOT, "include", "bbme.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bbme_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = C.CDLL(_capi.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, "declared in bbme.h but not exported: %s" % missing
    assert declared == set(_capi.SIGNATURES), (declared ^ set(_capi.SIGNATURES))
    assert _capi.lib().bbme_version().startswith(b"bbme")


def test_rccl_library_exports_its_header(bbme):
    """libbbme_rccl.so (the multi-GPU gather without torch) exports what include/bbme_rccl.h declares, and the sequence
    driver built on it exists and explains itself."""
    import subprocess
    from blockbasedmotionestimation_amd import build as _build
    header = open(os.path.join(ROOT, "include", "bbme_rccl.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bbme_[a-z0-9_]+)\s*\(", header))
    assert declared == {"bbme_gather_cells", "bbme_expand_gathered"}
    lib = C.CDLL(_build.RCCL_LIB)
    assert all(hasattr(lib, s) for s in declared)
    r = subprocess.run([_build.SEQ], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: bbme_seq" in r.stderr


def test_no_cpu_fallback_without_device(bbme):
    """The product fails loudly when no GPU can be used: there is no CPU compute path."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    z = np.zeros((128, 128), np.uint8)
    with pytest.raises(bbme.BbmeError) as e:
        bbme.MF(z, z, [30, 30], [16, 16])
    assert e.value.status == -5 and "no CPU fallback" in e.value.message


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "blockbasedmotionestimation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                for needle in ("from oracle", "import oracle", "bbme_oracle", "orc_", "oracle/_build", "oracle/_ref"):
                    assert needle not in text, "%s mentions %r" % (f, needle)


@pytest.mark.parametrize("w,h,blocks", [(584, 388, [16, 16, 16]), (1920, 1080, [16, 16, 16]), (3840, 2160, [16] * 4),
                                         (3840, 2160, [8] * 4), (2336, 1552, [32] * 4), (200, 120, [16, 16]),
                                         (640, 480, [16]), (100, 100, [8, 16, 8]), (333, 77, [4, 4])])
def test_padding_plan_matches_oracle(bbme, oracle, w, h, blocks):
    rc, pw, ph, px, py = oracle.plan_padding(w, h, blocks)
    if rc == 0:
        assert bbme.plan_padding(w, h, [b + 14 for b in blocks], blocks) == (pw, ph, px, py)
    else:
        with pytest.raises(bbme.BbmeError) as e:
            bbme.plan_padding(w, h, [b + 14 for b in blocks], blocks)
        assert e.value.status == {-1: -2, -2: -3}[rc]


def test_padding_plan_exhaustive_small(bbme, oracle):
    for blocks in ([16], [8, 8], [16, 16, 16], [4, 8]):
        for w in range(20, 150, 7):
            for h in (33, 64, 97, 128):
                rc, pw, ph, px, py = oracle.plan_padding(w, h, blocks)
                try:
                    got = bbme.plan_padding(w, h, [b + 2 for b in blocks], blocks)
                    assert rc == 0 and got == (pw, ph, px, py), (w, h, blocks)
                except bbme.BbmeError as e:
                    assert rc != 0 and e.status == {-1: -2, -2: -3}[rc], (w, h, blocks, rc, e.status)


def test_known_geometries(bbme):
    assert bbme.plan_padding(584, 388, [30] * 3, [16] * 3) == (640, 448, 28, 30)            # cfg1
    assert bbme.plan_padding(1920, 1080, [48] * 3, [16] * 3) == (1920, 1088, 0, 4)           # cfg2
    assert bbme.plan_padding(3840, 2160, [80] * 4, [16] * 4) == (3840, 2176, 0, 8)           # cfg3
    assert bbme.plan_padding(2336, 1552, [64] * 4, [32] * 4) == (2560, 1792, 112, 120)       # main_class.cpp:19-21


def test_pad_pyrdown_resize_match_oracle(bbme, oracle):
    rng = np.random.default_rng(3)
    for (h, w) in [(48, 64), (50, 70), (2, 2), (6, 4), (33, 31)]:
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(bbme.pad_zero(img, 5, 3), oracle.pad_zero(img, 5, 3))
        assert np.array_equal(bbme.pyr_down(img), oracle.pyr_down(img))
        assert np.array_equal(bbme.resize_x4(img), oracle.resize_linear_x4(img))
    # constant images stay constant, a pyrDown of a ramp stays a ramp in the interior
    c = np.full((32, 32), 201, np.uint8)
    assert np.all(bbme.pyr_down(c) == 201) and np.all(bbme.resize_x4(c) == 201)


def test_flow_class_codec_and_epe(bbme, oracle, tmp_path):
    flow = bbme.Flow()
    ref_file = os.path.join(GOLDEN, "flo_ramp_ref.flo")
    f = flow.ReadFlowFile(ref_file)
    assert np.array_equal(f, oracle.flo_read(ref_file))
    out = tmp_path / "w.flo"
    flow.WriteFlowFile(f, str(out))
    assert out.read_bytes() == open(ref_file, "rb").read()
    venus = flow.ReadFlowFile(os.path.join(GOLDEN, "gt_Venus_flow10.flo"))
    assert venus.shape == (380, 420, 2)
    rng = np.random.default_rng(9)
    est = venus + rng.normal(0, 1.5, venus.shape).astype(np.float32)
    est[5, 5] = 2e9                                              # estimate may be anything; GT decides "unknown"
    gt = venus.copy()
    gt[10:20, 10:30] = 1.666666752e9                             # Middlebury's unknown marker
    gt[3, 3, 1] = np.nan
    assert flow.CalculateMSE(gt, est) == oracle.calculate_mse(gt, est)
    assert flow.CalculateMSE(venus, venus) == 0.0
    sub = bbme.subsample_div4(np.arange(64 * 48 * 2, dtype=np.float32).reshape(48, 64, 2), 4, 8, 14, 8)
    assert np.array_equal(sub, oracle.subsample_div4(np.arange(64 * 48 * 2, dtype=np.float32).reshape(48, 64, 2), 4, 8, 14, 8))


def test_motion_to_color_equals_oracle(bbme, oracle, tmp_path, capsys):
    """Flow::MotionToColor / ShowImage of the product against the oracle (bit-exact) and against the image the
    reference's vendored colorcode.cpp made (tests/golden/color_ref.npz; one level in a handful of channels,
    see test_colour_coding_pinned_by_reference_colorcode)."""
    flow = bbme.Flow()
    ref = np.load(os.path.join(GOLDEN, "color_ref.npz"), allow_pickle=False)
    venus = flow.ReadFlowFile(os.path.join(GOLDEN, "gt_Venus_flow10.flo"))
    rng = np.random.default_rng(21)
    noisy = (rng.normal(0, 6, (97, 131, 2))).astype(np.float32)
    noisy[3, 4] = np.nan
    noisy[10:12] = 1.666666752e9
    noisy[20, 20] = (-0.0, 0.0)
    cases = [(venus, -1.0, "venus_auto"), (venus, 3.5, "venus_max3p5"), (ref["wheel_flow"], -1.0, "wheel_auto"),
             (ref["wheel_flow"], 40.0, "wheel_max40"), (noisy, -1.0, None), (noisy, 2.0, None),
             (np.zeros((4, 5, 2), np.float32), -1.0, None)]
    for f, maxmotion, key in cases:
        got = flow.MotionToColor(f, maxmotion, verbose=False)
        exp, exp_range = oracle.motion_to_color(f, maxmotion)
        assert got.dtype == np.uint8 and got.shape == f.shape[:2] + (3,)
        assert np.array_equal(got, exp)
        assert flow.last_range == exp_range
        if key is not None:
            d = np.abs(got.astype(np.int16) - ref[key].astype(np.int16))
            assert d.max() <= 1 and np.count_nonzero(d) <= 1e-4 * d.size
    flow.MotionToColor(venus, -1)
    assert capsys.readouterr().out == "max motion: 9.3750  motion range: u = -9.375 .. 7.000;  v = 0.000 .. 0.000\n"
    img = flow.MotionToColor(ref["wheel_flow"], verbose=False)
    out = tmp_path / "flowimg.ppm"
    flow.ShowImage(img, str(out))
    raw = out.read_bytes()
    head = b"P6\n181 121\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + img.size
    assert np.array_equal(np.frombuffer(raw[len(head):], np.uint8).reshape(img.shape), img[..., ::-1])
    with pytest.raises(bbme.BbmeError):
        flow.MotionToColor(np.zeros((4, 4, 3), np.float32))
    with pytest.raises(bbme.BbmeError):
        flow.ShowImage(np.zeros((4, 4), np.uint8), str(out))
    with pytest.raises(bbme.BbmeError) as e:
        flow.ShowImage(img, str(tmp_path / "no_such_dir" / "x.ppm"))
    assert e.value.status == -6


def test_async_flow_writer(bbme, tmp_path):
    """bbme_flo_writer_* (SURVEY 8f3): files written on the worker thread are byte for byte Flow::WriteFlowFile's, for a
    whole field and for the unpadded window of a padded one (main_class.cpp:63-70); I/O errors surface at wait()."""
    rng = np.random.default_rng(0)
    f = rng.standard_normal((40, 56, 2)).astype(np.float32)
    w = bbme.FlowWriter()
    w.submit(f, str(tmp_path / "a.flo"))
    w.submit(f, str(tmp_path / "b.flo"), pad_x=4, pad_y=6, width=40, height=20)
    w.wait()
    assert np.array_equal(bbme.Flow().ReadFlowFile(str(tmp_path / "a.flo")), f)
    bbme.Flow().WriteFlowFile(f[6:26, 4:44], str(tmp_path / "c.flo"))
    assert (tmp_path / "b.flo").read_bytes() == (tmp_path / "c.flo").read_bytes()
    with pytest.raises(bbme.BbmeError) as e:
        w.submit(f, str(tmp_path / "b.txt"))                        # extension check of WriteFlowFile (:147-152)
    assert e.value.status == -6
    w.submit(f, str(tmp_path / "no_such_dir" / "x.flo"))
    with pytest.raises(bbme.BbmeError) as e:
        w.wait()
    assert e.value.status == -6 and "problem writing" in e.value.message
    w.submit(f, str(tmp_path / "d.flo"))                            # usable again after an error
    w.wait()
    assert (tmp_path / "d.flo").read_bytes() == (tmp_path / "a.flo").read_bytes()
    w.close()


def test_flow_writer_pool_and_tickets(bbme, tmp_path):
    """bbme_flo_writer_create_pool / _ticket / _wait_ticket: several workers, one file each at a time; tickets count the jobs in
    submission order, wait(ticket) returns once every job up to it is on disk -- whatever the later ones are doing -- and a
    plain wait() covers everything.  What the sequence driver's round pipeline stands on (csrc/seq_schedule.hpp)."""
    rng = np.random.default_rng(3)
    cells = rng.integers(-50, 50, (240, 320, 2), dtype=np.int16)
    w = bbme.FlowWriter(workers=3)
    tickets = [w.submit_cells(cells, str(tmp_path / ("%02d.flo" % i))) for i in range(12)]
    assert tickets == list(range(1, 13))
    w.wait(tickets[3])                                             # jobs 1..4 are complete files now
    ref = None
    for i in range(4):
        data = (tmp_path / ("%02d.flo" % i)).read_bytes()
        assert len(data) == 12 + 8 * 640 * 480
        ref = ref or data
        assert data == ref
    w.wait(10 ** 9)                                                # a ticket beyond the last one = everything submitted
    w.wait()
    assert all((tmp_path / ("%02d.flo" % i)).read_bytes() == ref for i in range(12))
    # an I/O error of any worker surfaces at the next wait, and the pool stays usable
    w.submit_cells(cells, str(tmp_path / "no_such_dir" / "x.flo"))
    with pytest.raises(bbme.BbmeError) as e:
        w.wait()
    assert e.value.status == -6
    t = w.submit_cells(cells, str(tmp_path / "again.flo"))
    w.wait(t)
    assert (tmp_path / "again.flo").read_bytes() == ref
    w.close()
    with pytest.raises(bbme.BbmeError):
        bbme.FlowWriter(workers=0)


@pytest.mark.parametrize("pads", [(0, 0), (4, 6), (3, 5), (1, 0)])
@pytest.mark.parametrize("threads", ["1", "3"])
def test_async_flow_writer_from_cells(bbme, tmp_path, monkeypatch, pads, threads):
    """bbme_flo_writer_submit_cells: the worker expands the 2x2-cell grid (copy_to_all_pixels, motion_framework.cpp:815-826),
    strips the padding (main_class.cpp:63-70) and writes -- the file is byte for byte Flow::WriteFlowFile's of the dense
    field's window, for even and odd paddings, with one and several helper threads, over several bands of rows."""
    from blockbasedmotionestimation_amd.sequence import cells_to_words, expand_cells_host
    monkeypatch.setenv("BBME_WRITER_THREADS", threads)
    rng = np.random.default_rng(5)
    cells = rng.integers(-300, 300, (151, 160, 2), dtype=np.int16)          # 302 x 320 pixels: 2560-byte rows, several bands
    dense = expand_cells_host(cells_to_words(cells))
    px, py = pads
    wd, ht = 2 * cells.shape[1] - 2 * px, 2 * cells.shape[0] - 2 * py
    w = bbme.FlowWriter()
    w.submit_cells(cells, str(tmp_path / "cells.flo"), px, py, wd, ht)
    w.wait()
    bbme.Flow().WriteFlowFile(dense[py:py + ht, px:px + wd], str(tmp_path / "dense.flo"))
    assert (tmp_path / "cells.flo").read_bytes() == (tmp_path / "dense.flo").read_bytes()
    with pytest.raises(bbme.BbmeError) as e:                                  # window larger than the grid
        w.submit_cells(cells, str(tmp_path / "x.flo"), 2, 2, 2 * cells.shape[1], 10)
    assert e.value.status == -1
    with pytest.raises(bbme.BbmeError) as e:
        w.submit_cells(cells, str(tmp_path / "x.txt"), px, py, wd, ht)
    assert e.value.status == -6
    w.submit_cells(cells, str(tmp_path / "no_such_dir" / "x.flo"), px, py, wd, ht)
    with pytest.raises(bbme.BbmeError) as e:
        w.wait()
    assert e.value.status == -6
    w.close()


def test_flow_errors_raise_instead_of_exit(bbme, tmp_path):
    flow = bbme.Flow()
    good = open(os.path.join(GOLDEN, "flo_ramp_ref.flo"), "rb").read()
    for name, data in {"short.flo": good[:-1], "long.flo": good + b"x", "tag.flo": b"XXXX" + good[4:]}.items():
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(bbme.BbmeError) as e:
            flow.ReadFlowFile(str(p))
        assert e.value.status == -6
    for bad in (str(tmp_path / "nodot"), str(tmp_path / "x.txt"), None):
        with pytest.raises(bbme.BbmeError):
            flow.WriteFlowFile(np.zeros((2, 2, 2), np.float32), bad)
    with pytest.raises(bbme.BbmeError):
        flow.ReadFlowFile(None)
    with pytest.raises(bbme.BbmeError):
        flow.ReadFlowFile(str(tmp_path / "absent.flo"))


def test_parameter_validation(bbme):
    for search, block in [([30], [12]), ([30], [1]), ([30], [128]), ([0], [16]), ([400], [16])]:
        with pytest.raises(bbme.BbmeError):
            bbme.plan_padding(640, 480, search, block)
    with pytest.raises(ValueError):
        bbme.plan_padding(640, 480, [30, 30], [16])


def test_synth_pair_is_deterministic(bbme):
    a = bbme.synth_pair(96, 64, 77, max_motion=5)
    b = bbme.synth_pair(96, 64, 77, max_motion=5)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert a[0].dtype == np.uint8 and a[0].shape == (64, 96) and a[2].shape == (64, 96, 2)
    # noise-free pair: inside a motion tile, frame2 is frame1 moved by the tile's vector
    f1, f2, mo = bbme.synth_pair(96, 64, 78, max_motion=5, noise=0, tiles=1)
    dx, dy = (int(v) for v in mo[0, 0])
    ys, xs = np.mgrid[8:56, 8:88]
    assert np.array_equal(f2[ys + dy, xs + dx], f1[ys, xs])


def test_cli_builds_and_fails_loudly_without_gpu(bbme, tmp_path):
    """The C++ host side (MF / Flow classes + the reference driver as a CLI) links only the C-ABI."""
    import subprocess
    from blockbasedmotionestimation_amd import build as _build
    assert os.path.exists(_build.CLI)
    r = subprocess.run([_build.CLI], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    img = np.zeros((40, 48), np.uint8)
    for name in ("a.pgm", "b.pgm"):
        with open(tmp_path / name, "wb") as f:
            f.write(b"P5\n48 40\n255\n" + img.tobytes())
    r = subprocess.run([_build.CLI, str(tmp_path / "a.pgm"), str(tmp_path / "missing.pgm")], capture_output=True, text=True)
    assert r.returncode == 1 and "Could not open one of the images" in r.stderr
    if not os.path.exists("/dev/kfd"):
        r = subprocess.run([_build.CLI, str(tmp_path / "a.pgm"), str(tmp_path / "b.pgm"), "--levels", "1", "--block", "16",
                            "--search", "30"], capture_output=True, text=True)
        assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.parametrize("search,block", [(30, 16), (48, 16), (80, 16), (72, 8), (64, 32), (16, 16), (17, 16), (21, 16),
                                          (10, 16), (142, 16), (33, 8)])
def test_spiral_table_equals_reference_loop_walk(bbme, oracle, search, block):
    """The product's rank -> (dx, dy) table (built in libbbme.so) against the oracle's literal walk of
    motion_framework.cpp:326-411."""
    from blockbasedmotionestimation_amd import _capi
    n = C.c_int()
    _capi.check(_capi.lib().bbme_spiral_host(search, block, None, None, 0, C.byref(n)))
    dx = np.zeros(n.value, np.int16)
    dy = np.zeros(n.value, np.int16)
    _capi.check(_capi.lib().bbme_spiral_host(search, block, dx.ctypes.data, dy.ctypes.data, n.value, C.byref(n)))
    odx, ody = oracle.spiral_walk(search - block)
    assert n.value == len(odx)
    assert np.array_equal(dx, odx) and np.array_equal(dy, ody)


@pytest.mark.parametrize("waves", [1, 2])
@pytest.mark.parametrize("block", [8, 16, 32])
@pytest.mark.parametrize("rng", [0, 1, 2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 45, 63])
def test_search_plan_covers_every_candidate_once(bbme, rng, block, waves):
    """k_search_fast's work split: every (column group, candidate row) in exactly one strip, strips of a
    round all of the round's height, at most 64 per wave and round, rounds full except possibly the last -- for one wave
    per macroblock and for the two waves that share a block on levels of few blocks."""
    from blockbasedmotionestimation_amd import _capi
    cap = 256
    lanes = 64 * waves
    rounds = np.zeros(cap, np.uint32)
    tasks = np.zeros((cap, lanes), np.uint32)
    nr, groups, pitch = C.c_int(), C.c_int(), C.c_int()
    _capi.check(_capi.lib().bbme_search_plan_host_waves(rng, block, waves, rounds.ctypes.data, cap, C.byref(nr), tasks.ctypes.data,
                                                        C.byref(groups), C.byref(pitch)))
    if waves == 1:                               # the one-wave entry point is the same plan
        r1, t1, n1 = np.zeros(cap, np.uint32), np.zeros((cap, 64), np.uint32), C.c_int()
        _capi.check(_capi.lib().bbme_search_plan_host(rng, block, r1.ctypes.data, cap, C.byref(n1), t1.ctypes.data, None, None))
        assert n1.value == nr.value and np.array_equal(r1, rounds) and np.array_equal(t1, tasks)
    n = 2 * rng + 1
    assert groups.value == (n + 3) // 4 and 1 <= nr.value <= cap
    assert pitch.value % 2 == 1 and pitch.value >= groups.value + block // 4
    # every candidate (dx index, dy index) of the (2R+1)^2 square exactly once.  rounds[] = S | kind << 8: kind 0 strips of a
    # column group (four candidate columns x S rows); kind 1 one candidate row of a group by the four lanes of a quad, a
    # quarter of the block's rows each; kind 2 the last candidate column (dx = +R), one candidate per lane (tight plan, n = 4G+1)
    covered = np.zeros((4 * groups.value, n), np.int32)
    cost = 0.0
    for r in range(nr.value):
        s, kind = int(rounds[r]) & 0xFF, int(rounds[r]) >> 8
        assert kind in (0, 1, 2)
        busy = 0
        if kind == 0:
            assert s in ((8, 4, 2, 1) if block == 32 or waves == 2 else (16, 8, 4, 2, 1))
            cost += s
        else:
            assert block <= 16 and n % 4 == 1 and s == 1
            cost += 0.25
        quads = {}
        for lane, t in enumerate(tasks[r]):
            if t == 0xFFFFFFFF:
                continue
            busy += 1
            if kind == 0:
                g, dy0 = int(t & 0xFF), int((t >> 8) & 0xFF)
                assert g < groups.value and dy0 + s <= n
                covered[4 * g:4 * g + 4, dy0:dy0 + s] += 1
            elif kind == 1:
                g, dyi, part = int(t & 0xFF), int((t >> 8) & 0xFF), int((t >> 16) & 3)
                assert part == lane % 4 and g < groups.value - 1 and dyi < n
                quads.setdefault((lane // 4, g, dyi), []).append(part)
            else:
                covered[n - 1, int(t & 0xFF)] += 1
        for (_, g, dyi), parts in quads.items():
            assert parts == [0, 1, 2, 3]
            covered[4 * g:4 * g + 4, dyi] += 1
        assert 1 <= busy <= lanes
        if r < nr.value - 1 and kind == 0 and (int(rounds[r + 1]) >> 8) == 0:
            assert busy == lanes or s == 1
    assert np.all(covered[:n] == 1)
    # a round costs its strip height (rim rounds a quarter of a row); the plan should stay close to the ideal n * n / 4 / lanes
    assert cost <= -(-n * groups.value // lanes) + 2
    if n % 4 == 1 and n >= 9 and block <= 16:
        assert cost <= n * n / 4.0 / lanes + 1.6
