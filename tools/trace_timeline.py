#!/usr/bin/env python3
"""Timeline of a window of a rocprofv3 kernel_trace.csv: start offset, duration, queue, grid, kernel.
    python tools/trace_timeline.py trace.csv <anchor-kernel-substring> <occurrence> <count>
The window starts at the <occurrence>-th dispatch (0-based) whose name contains the anchor and shows <count> dispatches
in start order."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor, occ, count = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = hits[occ]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void gpn::", "").replace("gpn::", "")[:48]
    print("%10.1f us  +%8.1f us  q%-3s grid=%7d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"),
                                                      int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), name))
