#!/usr/bin/env python3
"""Per-launch SQ counters of the plain search launches collected by scripts/pmc_search.sh, with the derived figures the
roofline discussion needs (quad-cycle counters: SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*, MI355X_MICROARCH.md).

    python scripts/pmc_search_report.py OUTDIR [--json profiles/rNN_pmc_search_sq.json]"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--match", default="k_search_fast")
ap.add_argument("--json")
a = ap.parse_args()
acc = defaultdict(lambda: defaultdict(list))       # (kernel, grid, ...) -> counter -> values (one per dispatch)
for g in sorted(glob.glob(os.path.join(a.dir, "g*"))):
    if not os.path.isdir(g):
        continue
    per = defaultdict(lambda: defaultdict(float))
    name = {}
    for f in glob.glob(os.path.join(g, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if a.match not in r["Kernel_Name"]:
                continue
            d = int(r["Dispatch_Id"])
            per[d][r["Counter_Name"]] += float(r["Counter_Value"])
            short = r["Kernel_Name"].split("(")[0].replace("void bbme::", "")
            name[d] = (short, int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0),
                       int(r.get("VGPR_Count", r.get("Arch_VGPR_Count", 0)) or 0), int(r.get("SGPR_Count", 0) or 0),
                       int(r.get("LDS_Block_Size", 0) or 0))
    for d, cs in per.items():
        for k, v in cs.items():
            acc[name[d]][k].append(v)
out = {}
for key in sorted(acc, key=lambda k: -k[1]):
    c = {k: sum(v) / len(v) for k, v in acc[key].items()}
    short, grid, wg, vgpr, sgpr, lds = key
    waves = c.get("SQ_WAVES", 0)
    d = {"launches": len(next(iter(acc[key].values()))), "grid_threads": grid, "workgroup": wg, "vgpr": vgpr, "sgpr": sgpr,
         "lds_bytes": lds, "counters": c}
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        d["per_wave_quadcycles"] = wc / max(waves, 1)
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
                  "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC"):
            if k in c:
                d["share_of_wave_cycles:" + k] = c[k] / wc
    if waves:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM"):
            if k in c:
                d["per_wave:" + k] = c[k] / waves
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_share_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    if "SQ_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        d["sq_busy_over_gui_active"] = c["SQ_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"]
    out["%s grid=%d" % (short, grid)] = d
# the figures bench.py quotes for the level-0 launch (the largest grid): VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles, summed over
# waves) x 4 / 1024 SIMDs against the launch's GRBM_GUI_ACTIVE / 8 XCDs
import hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "blockbasedmotionestimation_amd", "csrc", "bbme_kernels.hpp"), "rb").read()
big = max(out.values(), key=lambda v: v["grid_threads"]) if out else None
if big and "SQ_ACTIVE_INST_VALU" in big["counters"] and "GRBM_GUI_ACTIVE" in big["counters"]:
    c = big["counters"]
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    out["summary_level0_launch"] = {
        "kernel_cycles_per_xcd": cyc,
        "valu_busy": c["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc,
        "valu_insts_per_wave": big.get("per_wave:SQ_INSTS_VALU"), "salu_insts_per_wave": big.get("per_wave:SQ_INSTS_SALU"),
        "lds_insts_per_wave": big.get("per_wave:SQ_INSTS_LDS"), "vmem_insts_per_wave": big.get("per_wave:SQ_INSTS_VMEM"),
        "lds_bank_conflict_share_of_lds_cycles": big.get("lds_bank_conflict_share_of_lds_cycles"),
        "wave_cycles_waiting_to_issue": big.get("share_of_wave_cycles:SQ_WAIT_INST_ANY"),
        "wave_cycles_waiting_on_lds_issue": big.get("share_of_wave_cycles:SQ_WAIT_INST_LDS"),
        "wave_cycles_parked": big.get("share_of_wave_cycles:SQ_WAIT_ANY"),
        "waves": c.get("SQ_WAVES"), "mean_waves_per_simd": c.get("SQ_WAVE_CYCLES", 0) * 4.0 / 1024.0 / cyc,
    }
out["kernel_source_sha256"] = hashlib.sha256(src).hexdigest()
print(json.dumps(out, indent=1, sort_keys=True))
if a.json:
    json.dump(out, open(a.json, "w"), indent=1, sort_keys=True)
