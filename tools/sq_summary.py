#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass of SQ / GRBM counters into per-kernel sums and the derived figures DESIGN.md quotes:
   mfma_busy_frac   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   (MFMA-pipe busy cycles summed over the
                      chip's 1024 SIMDs, against the shader cycles the dispatches lasted: GRBM_GUI_ACTIVE sums 8 XCDs)
   f64_mfma_flops   = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512                     (the counter ticks in units of 512 flops)
   eff_clock_GHz    = GRBM_GUI_ACTIVE / 8 / kernel time                     (MI355X_MICROARCH.md, DVFS give-back)
   python3 tools/sq_summary.py <pmc dir> out.json N B calls"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import csrc_digest  # noqa: E402


def short(name):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    d, out, N, B, calls = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    files = glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection csv under {d}")
    per, seen = {}, {}
    for f in files:
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            e = per.setdefault(k, {"launches": 0, "ns": 0.0, "counters": {}})
            disp = (k, row.get("Dispatch_Id"))
            if disp not in seen:
                seen[disp] = True
                e["launches"] += 1
                if row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    e["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            c = e["counters"]
            c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k, e in per.items():
        c = e["counters"]
        gui, mfma = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        e["derived"] = {}
        if gui > 0:
            e["derived"]["mfma_busy_frac_of_simd_cycles"] = mfma / (gui / 8.0 * 1024.0)
        if c.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
            e["derived"]["f64_mfma_flops"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
            if e["ns"] > 0:
                e["derived"]["f64_mfma_tflops_in_kernel_time"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0 / e["ns"] / 1e3
        if c.get("SQ_WAVE_CYCLES"):
            w = c["SQ_WAVE_CYCLES"]
            e["derived"]["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / w
            e["derived"]["wait_inst_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / w
            e["derived"]["active_inst_any_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / w
        if e["ns"] > 0 and c.get("GRBM_GUI_ACTIVE"):
            e["derived"]["eff_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / e["ns"]
        e["ms_total"] = e.pop("ns") / 1e6
    res = {"meta": {"N": N, "B": B, "calls_profiled": calls, "csrc_digest": csrc_digest(),
                    "command": "tools/sq_counters.sh (one rocprofv3 --pmc pass, counters only) on tools/profile_mll.py %d %d %d"
                               % (N, B, calls - 1),
                    "note": "profiled passes run at a lower clock than un-profiled ones; compare fractions, not times"},
           "kernels": per}
    json.dump(res, open(out, "w"), indent=1)
    for k in sorted(per, key=lambda k: -per[k]["ms_total"])[:6]:
        print(k, per[k]["launches"], "launches", round(per[k]["ms_total"], 3), "ms", per[k]["derived"])


if __name__ == "__main__":
    main()
