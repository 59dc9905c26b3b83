#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build container:

    python tests/golden/make_golden.py

1. hotpath_*.npz -- inputs (level planes AFTER padding + pyramid) and expected outputs of the hot
   path (MV grid after the search of every level and after every regulariser sweep, final dense
   flow) for small seeded cases.  Produced by the CPU oracle (oracle/bbme_oracle.c), because the
   reference's core cannot be built here (needs OpenCV) and holds no vectors of its own:
   regression vectors, PARITY UNPINNED with respect to the reference binary.
2. flo_ramp_ref.flo -- a 7x5 .flo written by the REFERENCE's own Middlebury code
   (middlebury/flow-code/flowIO.cpp, compiled into oracle/_ref/flo_ref): pins the codec.
3. gt_stats.json -- known answers computed with the reference's own reader on the 8 ground-truth
   files it ships (size, unknown-pixel count, sums, sha256), plus gt_Venus_flow10.flo copied as
   data (smallest GT file) so the GPU box can run the known-answer test without /root/reference.
4. color_ref.npz -- colour coding made by the REFERENCE's own vendored colour-wheel code
   (middlebury/flow-code/colorcode.cpp computeColor, compiled into oracle/_ref/flo_ref, called by
   oracle/ref_flo_driver.cpp): Venus ground truth with the automatic radius and with maxmotion 3.5
   (exercises the out-of-range branch), and a synthetic wheel field (all angles, radii 0..1.5,
   some unknown pixels).  gt_stats.json also gets the sha256 of that output for all 8 GT files.
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import bbme_oracle as O                                     # noqa: E402
from blockbasedmotionestimation_amd.synth import synth_pair             # noqa: E402
from helpers import oracle_schedule                                     # noqa: E402

REF = "/root/reference"

CASES = {
    # name: (w, h, search, block, seed, max_motion)
    "hotpath_b16_r7_l3": (160, 112, [30, 30, 30], [16, 16, 16], 2001, 6),
    "hotpath_b16_r16_l2": (128, 96, [48, 48], [16, 16], 2002, 12),
    "hotpath_b8_r32_l2": (96, 64, [72, 72], [8, 8], 2003, 10),
    "hotpath_b32_r16_l2": (256, 128, [64, 64], [32, 32], 2004, 14),
    "hotpath_mixed_l3": (192, 128, [24, 40, 30], [8, 16, 8], 2005, 8),
    # the reference's second literal set (main_class.cpp:15-17): block {16,16,32}, search {32,32,42}; 376 x 250 pads to 384 x 256
    "hotpath_ref2_l3": (376, 250, [32, 32, 42], [16, 16, 32], 2006, 8),
    # 2 x 2 blocks as a level's own block size (level 0 here), under 4 x 4
    "hotpath_block2_l2": (96, 64, [12, 16], [2, 4], 2007, 5),
}


# the variants of SURVEY 8(f4): "variant_*.npz", made with `python tests/golden/make_golden.py variants` (the files above
# are not rewritten).  raster: MF::find_min_block (:246-294) as the search; jacobi: NOT the reference -- the product's
# opt-in fast regulariser as the oracle defines it.
VARIANTS = {
    "variant_raster_b16_r7_l3": (160, 112, [30, 30, 30], [16, 16, 16], 2101, 6, "raster"),
    "variant_raster_b8_r32_l2": (96, 64, [72, 72], [8, 8], 2102, 10, "raster"),
    "variant_jacobi_b16_r7_l3": (160, 112, [30, 30, 30], [16, 16, 16], 2103, 6, "jacobi"),
}


def make_hotpath(name, w, h, search, block, seed, mm, mode=None):
    f1, f2, _ = synth_pair(w, h, seed, max_motion=mm)
    L = len(block)
    omf = O.OracleMF(f1, f2, search, block)
    if mode == "raster":
        omf.set_raster_search(True)
    if mode == "jacobi":
        omf.set_jacobi_regularizer(True)
    data = {"search_size": np.array(search, np.int32), "block_size": np.array(block, np.int32),
            "frame1": f1, "frame2": f2,
            "geometry": np.array([omf.padded_width, omf.padded_height, omf.padding_x, omf.padding_y], np.int32)}
    for lvl in range(L):
        data["plane1_l%d" % lvl] = omf.image(lvl, 1).copy()
        data["plane2_l%d" % lvl] = omf.image(lvl, 2).copy()
    stages = []

    def on_stage(kind, lvl, b, mvs):
        key = "mv_%02d_%s_l%d_b%d" % (len(stages), kind, lvl, b)
        stages.append(key)
        data[key] = mvs.astype(np.int16)

    data["flow"] = oracle_schedule(omf, L, on_stage)
    data["stages"] = np.array(stages)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "stages:", len(stages), "flow", data["flow"].shape)


def wheel_field():
    """All angles and radii 0 .. 1.5 of the normalising radius, a band of unknown pixels, signed zeros."""
    y, x = np.mgrid[-60:61, -90:91].astype(np.float32)
    f = np.stack([x * np.float32(0.37), y * np.float32(0.53)], -1).astype(np.float32)
    f[5:9, :, 0] = 1e10                                    # unknown (rw_flow.cpp:39-43)
    f[60, :, 1] = -0.0                                     # atan2 of a negative zero: the +pi / -pi seam
    return f


def color_by_reference(flo_ref, flo_path, shape, maxmotion=None):
    out = os.path.join("/tmp", "bbme_color_ref.bgr")
    cmd = [flo_ref, "color", flo_path, out] + ([repr(maxmotion)] if maxmotion else [])
    subprocess.check_call(cmd)
    return np.fromfile(out, np.uint8).reshape(shape[0], shape[1], 3)


def main():
    O.build(force=True)
    if sys.argv[1:] == ["variants"]:
        for name, cfg in VARIANTS.items():
            make_hotpath(name, *cfg)
        return
    if len(sys.argv) == 3 and sys.argv[1] == "only":          # one hot-path case; the other files are not rewritten
        make_hotpath(sys.argv[2], *{**CASES, **VARIANTS}[sys.argv[2]])
        return
    for name, cfg in CASES.items():
        make_hotpath(name, *cfg)
    flo_ref = O.FLO_REF
    assert os.path.exists(flo_ref), "oracle/_ref/flo_ref missing (needs /root/reference)"
    subprocess.check_call([flo_ref, "ramp", "7", "5", os.path.join(HERE, "flo_ramp_ref.flo")])
    stats = {}
    gt_dir = os.path.join(REF, "middlebury", "gt-flow")
    for seq in sorted(os.listdir(gt_dir)):
        p = os.path.join(gt_dir, seq, "flow10.flo")
        w, h, unk, su, sv = subprocess.check_output([flo_ref, "stats", p]).decode().split()
        stats[seq] = {"width": int(w), "height": int(h), "unknown": int(unk), "sum_u": float(su), "sum_v": float(sv),
                      "bytes": os.path.getsize(p), "sha256": hashlib.sha256(open(p, "rb").read()).hexdigest()}
        shape = (int(h), int(w))
        stats[seq]["color_sha256"] = hashlib.sha256(color_by_reference(flo_ref, p, shape).tobytes()).hexdigest()
    json.dump(stats, open(os.path.join(HERE, "gt_stats.json"), "w"), indent=1, sort_keys=True)
    venus = os.path.join(gt_dir, "Venus", "flow10.flo")
    vshape = (stats["Venus"]["height"], stats["Venus"]["width"])
    wheel = wheel_field()
    wheel_path = "/tmp/bbme_wheel.flo"
    O.flo_write(wheel_path, wheel)
    np.savez_compressed(os.path.join(HERE, "color_ref.npz"),
                        venus_auto=color_by_reference(flo_ref, venus, vshape),
                        venus_max3p5=color_by_reference(flo_ref, venus, vshape, 3.5),
                        wheel_flow=wheel,
                        wheel_auto=color_by_reference(flo_ref, wheel_path, wheel.shape),
                        wheel_max40=color_by_reference(flo_ref, wheel_path, wheel.shape, 40.0))
    shutil.copyfile(os.path.join(gt_dir, "Venus", "flow10.flo"), os.path.join(HERE, "gt_Venus_flow10.flo"))
    os.chmod(os.path.join(HERE, "gt_Venus_flow10.flo"), 0o644)
    print("gt stats:", {k: (v["width"], v["height"], v["unknown"]) for k, v in stats.items()})


if __name__ == "__main__":
    main()
