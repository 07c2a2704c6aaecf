#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv: time per (kernel, grid) bucket."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nevals = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    short = n.split("(")[0].replace("void gpn::", "").replace("gpn::", "")[:60]
    d[(short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append(dur)
tot = sum(sum(v) for v in d.values())
print("total kernel time %.3f ms per eval" % (tot / 1e3 / nevals))
for k in sorted(d, key=lambda k: -sum(d[k]))[:40]:
    v = d[k]
    print("%-62s grid=%6d  n/eval=%7.1f  per-eval %8.1f us  avg %8.1f us" % (k[0], k[1], len(v) / nevals, sum(v) / nevals, sum(v) / len(v)))
