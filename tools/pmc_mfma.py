#!/usr/bin/env python3
"""MFMA utilisation per launch from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass
(north_star: "MFMA utilisation on the SYRK update reported from rocprof against gfx950 peak").
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md), SQ_VALU_MFMA_BUSY_CYCLES over all
1024 SIMDs:  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: pmc_mfma.py <counter_collection.csv> <out.json> <workload>"""
import collections, csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    key = (r["Dispatch_Id"], r["Kernel_Name"], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X", ""))
    by[key][r["Counter_Name"]] = float(r["Counter_Value"])
out = []
for (did, name, grid), c in by.items():
    if "gemm_nt_kernel" not in name or "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    active = c["GRBM_GUI_ACTIVE"] / 8.0
    if active <= 0:
        continue
    out.append({"kernel": name.split("(")[0][-60:], "grid": grid, "gui_active_cycles_per_xcd": active,
                "mfma_busy_cycles": c["SQ_VALU_MFMA_BUSY_CYCLES"], "mfma_utilisation": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (active * 1024.0)})
out.sort(key=lambda d: -d["gui_active_cycles_per_xcd"])
tot_busy = sum(d["mfma_busy_cycles"] for d in out)
tot_act = sum(d["gui_active_cycles_per_xcd"] for d in out)
res = {"workload": sys.argv[3], "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; utilisation = BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)",
       "all_contraction_launches": {"launches": len(out), "mfma_utilisation": tot_busy / (tot_act * 1024.0) if tot_act else None},
       "largest_launches": out[:8]}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(json.dumps(res["all_contraction_launches"]), [round(d["mfma_utilisation"], 3) for d in out[:8]])
