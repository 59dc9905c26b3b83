#!/usr/bin/env python3
"""Turns the counter CSVs of the two rocprofv3 --pmc passes into per-launch HBM traffic.

    python scripts/pmc_report.py --fetch OUT/fetch --write OUT/write --calib-mib 1024 --out profiles/r01_pmc_search.json

FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B per count as rocprofv3 reports them
(MI355X_MICROARCH.md, HBM: hbm_bytes = counter * 1024); on gfx950 FETCH_SIZE under-counts by a
factor that depends on the access shape, so it is scaled by the factor measured on the calibration
kernel of the same run (known bytes / counted bytes).  WRITE_SIZE is taken as is."""
import argparse
import csv
import glob
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(dirpath, counter):
    rows = []
    for f in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), float(r["Counter_Value"])))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--calib-mib", type=int, default=1024)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    fetch = load(a.fetch, "FETCH_SIZE")
    write = load(a.write, "WRITE_SIZE")
    calib = [v for n, g, v in fetch if "k_calib_read_dword" in n]
    # the first calibration launch reads cold HBM; use the last one as well and report both
    known = a.calib_mib * 1024 * 1024
    scale = [known / (v * 1024.0) for v in calib]
    out = {"units": "bytes per launch", "fetch_counter_unit_bytes": 1024,
           "calibration": {"known_bytes": known, "fetch_counts": calib, "scale_known_over_counted": scale},
           "kernels": {}}
    use_scale = scale[0] if scale else 1.0
    agg = defaultdict(lambda: defaultdict(list))
    for n, g, v in fetch:
        agg[(n, g)]["fetch"].append(v * 1024.0)
    for n, g, v in write:
        agg[(n, g)]["write"].append(v * 1024.0)
    for (n, g), d in sorted(agg.items()):
        if "k_search" not in n and "k_expand" not in n and "k_calib" not in n:
            continue
        f = sum(d["fetch"]) / max(len(d["fetch"]), 1)
        w = sum(d["write"]) / max(len(d["write"]), 1)
        out["kernels"]["%s grid=%d" % (n.split("(")[0].replace("void bbme::", ""), g)] = {
            "launches": len(d["fetch"]), "fetch_bytes_raw": f, "fetch_bytes_scaled": f * use_scale, "write_bytes": w,
            "hbm_bytes": f * use_scale + w}
    src = open(os.path.join(ROOT, "blockbasedmotionestimation_amd", "csrc", "bbme_kernels.hpp"), "rb").read()
    out["kernel_source_sha256"] = hashlib.sha256(src).hexdigest()
    json.dump(out, open(a.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
