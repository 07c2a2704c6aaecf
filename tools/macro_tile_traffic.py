#!/usr/bin/env python3
"""Post-process tools/macro_tile_ab.sh: HBM-side bytes per launch of the contraction kernel from the separate FETCH_SIZE /
WRITE_SIZE passes (KB -> B; FETCH x 2: gfx950 wide-read correction, as tools/pmc_traffic.py) per variant, next to the
timings of tools/gemm_ab.py.  usage: macro_tile_traffic.py <dir> <out.json>"""
import csv, glob, json, os, re, sys
d, out = sys.argv[1], sys.argv[2]
M, K = 30720, 2048
alg = {"operands_once": M * K * 8.0, "c_lower_read_plus_write": 2 * 8.0 * M * (M + 1) / 2}
res = {"shape": "M = 30720 lower-tile, K = 2048, C -= A A^T (C3's first trailing update)", "algorithmic_bytes": sum(alg.values()),
       "algorithmic_bytes_parts": alg, "flops": float(M) * (M + 1) * K, "variants": {}}
names = {11: "128x128 tile, 8 waves of 32x64, pipelined K loop (shipped)", 12: "128x128 tile, 8 waves of 32x64, plain K loop",
         13: "256x128 macro tile, 16 waves of 32x64, ONE workgroup / CU, 96 KB LDS, plain K loop"}
for v in (11, 12, 13):
    e = {"kernel": names[v]}
    for ctr, key, mul in (("FETCH_SIZE", "fetch_bytes_per_launch_corrected_x2", 2048.0), ("WRITE_SIZE", "write_bytes_per_launch", 1024.0)):
        tot, n = 0.0, 0
        for f in glob.glob(os.path.join(d, "pmc_%s_v%d" % (ctr.split("_")[0].lower(), v), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr and "gemm_nt_kernel" in r["Kernel_Name"]:
                    tot += float(r["Counter_Value"]); n += 1
        e[key] = tot * mul / n if n else None
        e[key + "_launches"] = n
    if e.get("fetch_bytes_per_launch_corrected_x2") and e.get("write_bytes_per_launch"):
        e["bytes_per_launch"] = e["fetch_bytes_per_launch_corrected_x2"] + e["write_bytes_per_launch"]
        e["traffic_over_algorithmic"] = e["bytes_per_launch"] / res["algorithmic_bytes"]
    res["variants"][str(v)] = e
ab = os.path.join(d, "gemm_ab.txt")
if os.path.exists(ab):
    res["timing_same_box"] = [l.strip() for l in open(ab) if l.startswith("M=")]
    for l in res["timing_same_box"]:
        if "M= 30720" in l and "lower=1" in l:
            for v, ms, tf in re.findall(r"v(\d+)\s+([\d.]+) ms\s+([\d.]+) TF", l):
                if v in res["variants"]:
                    res["variants"][v]["ms"] = float(ms); res["variants"][v]["tflops"] = float(tf) / 1e0
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
