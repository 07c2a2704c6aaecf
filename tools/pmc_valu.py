#!/usr/bin/env python3
"""VALU-bound evidence for the HBM-side kernels (K assembly, gradient sweep) from a rocprofv3 pass with
--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (counters only).  A fp64 vector
instruction occupies its SIMD for 4 cycles per wave, so with I = VALU wave-instructions of a launch the
issue floor is  I * 4 / 1024 SIMDs  cycles; the launch took GRBM_GUI_ACTIVE / 8 cycles (the counter sums
the 8 XCDs).  valu_issue_fraction = floor / duration: how much of the launch is explained by VALU issue
alone (independent of any HBM traffic).
usage: pmc_valu.py <counter_collection.csv> <out.json> <tag>"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    by[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for (did, name), c in by.items():
    short = "kmat_kernel" if "kmat_kernel" in name else "grad_sweep_kernel" if "grad_sweep" in name else None
    if short is None or "SQ_INSTS_VALU" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    a = agg[short]
    a["launches"] += 1
    a["valu_wave_instructions"] += c["SQ_INSTS_VALU"]
    a["cycles"] += c["GRBM_GUI_ACTIVE"] / 8.0
    a["active_inst_valu"] += c.get("SQ_ACTIVE_INST_VALU", 0.0)
    a["sq_busy_cycles"] += c.get("SQ_BUSY_CYCLES", 0.0)
out = {"tag": sys.argv[3], "method": __doc__.split("usage")[0].strip(), "kernels": {}}
for k, a in agg.items():
    floor = a["valu_wave_instructions"] * 4.0 / 1024.0
    out["kernels"][k] = {"launches": int(a["launches"]), "valu_wave_instructions_per_launch": a["valu_wave_instructions"] / a["launches"],
                         "cycles_per_launch": a["cycles"] / a["launches"], "valu_issue_floor_cycles_per_launch": floor / a["launches"],
                         "valu_issue_fraction": floor / a["cycles"] if a["cycles"] else None,
                         "sq_active_inst_valu_per_launch": a["active_inst_valu"] / a["launches"]}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
