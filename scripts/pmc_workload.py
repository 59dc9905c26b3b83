#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a calibration read of known size followed by a few
whole-pyramid estimates of a bench workload.  Run it under rocprofv3 once per counter group:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python3 scripts/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python3 scripts/pmc_workload.py

and feed the two counter CSVs to scripts/pmc_report.py."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg3")
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--calib-mib", type=int, default=1024)
a = ap.parse_args()

import blockbasedmotionestimation_amd as bbme                    # noqa: E402
from blockbasedmotionestimation_amd import _capi                 # noqa: E402
from bench import WORKLOADS                                      # noqa: E402

_capi.check(_capi.lib().bbme_calibrate_read(0, a.calib_mib, 2))
w, h, search, block, levels, _ = WORKLOADS[a.workload]
f1, f2, _ = bbme.synth_pair(w, h, 1030, max_motion=24)
mf = bbme.MF(f1, f2, [search] * levels, [block] * levels, levels)
mf.set_speculation(False)        # one plain search launch per level, as in the eager pass bench.py's roofline comes from
for _ in range(a.iters):
    mf.estimate_async()
    mf.synchronize()
mf.close()
print("pmc workload done: calib %d MiB x2, %d estimates of %s" % (a.calib_mib, a.iters, a.workload))
