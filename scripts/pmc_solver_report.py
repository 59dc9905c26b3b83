#!/usr/bin/env python3
"""Prints, per dispatch of the kernels matching --match, the counters collected by scripts/pmc_solver.sh."""
import argparse, csv, glob, os
from collections import defaultdict
ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--match", default="k_reg_solve")
a = ap.parse_args()
rows = defaultdict(dict)          # (group, dispatch id) -> {counter: value}
names = {}
for g in sorted(glob.glob(os.path.join(a.dir, "g*"))):
    for f in glob.glob(os.path.join(g, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if a.match not in r["Kernel_Name"]:
                continue
            key = (os.path.basename(g), int(r["Dispatch_Id"]))
            rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            names[key] = r["Kernel_Name"].split("(")[0][-24:] + " grid=" + r.get("Grid_Size", "?")
for key in sorted(rows):
    print(key[0], key[1], names[key], " ".join("%s=%g" % (k.replace("_sum", ""), v) for k, v in sorted(rows[key].items())))
