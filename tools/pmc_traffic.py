#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md 'HBM' prescribes) of `bench.py --no-extras` into
HBM bytes per launch of the contraction kernel.  gfx950 corrections: the counters are in KB;
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -- the
kernel's operand traffic is LDS-DMA dwordx4, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <workload>"""
import collections, csv, json, sys

def per_grid(path, counter, match):
    """{workgroups per launch: [launches, summed counter]} of the kernels whose name contains `match`"""
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and match in r["Kernel_Name"]:
            g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)))
            wgs = int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0)
            if not wgs:        # the big tile runs as 8 waves (128, 128, 32, 64), every other variant as 4
                wgs = 512 if "<128, 128, 32, 64" in r["Kernel_Name"] or "<128, 128, 64, 32" in r["Kernel_Name"] else 256
            if g % wgs:
                continue
            g //= wgs
            d[g][0] += 1
            d[g][1] += float(r["Counter_Value"])
    return d


def per_kernel(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"]
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
    return d

fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
nf = sum(v[0] for k, v in fetch.items() if "gemm_nt_kernel" in k)
bf = sum(v[1] for k, v in fetch.items() if "gemm_nt_kernel" in k) * 1024.0 * 2.0
nw = sum(v[0] for k, v in write.items() if "gemm_nt_kernel" in k)
bw = sum(v[1] for k, v in write.items() if "gemm_nt_kernel" in k) * 1024.0
kf = sum(v[1] for k, v in fetch.items() if "kmat_kernel" in k) * 1024.0      # scalar-ish 8 B/lane reads of X: no x2
kfn = sum(v[0] for k, v in fetch.items() if "kmat_kernel" in k)
kw = sum(v[1] for k, v in write.items() if "kmat_kernel" in k) * 1024.0
kwn = sum(v[0] for k, v in write.items() if "kmat_kernel" in k)
out = {"workload": sys.argv[4], "kernel": "gemm_nt_kernel (all variants)", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "kmat_bytes_per_launch": (kf / max(kfn, 1) + kw / max(kwn, 1)) if kfn and kwn else None,
       "kmat_fetch_bytes_per_launch": kf / max(kfn, 1) if kfn else None, "kmat_write_bytes_per_launch": kw / max(kwn, 1) if kwn else None,
       "fetch_bytes_per_launch_corrected_x2": bf / max(nf, 1), "write_bytes_per_launch": bw / max(nw, 1),
       "gemm_bytes_per_launch": bf / max(nf, 1) + bw / max(nw, 1),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --no-extras --no-cpu-baseline`; KB->B; FETCH x2 (gfx950 wide-read correction)"}
# the SYRK trailing updates alone (bench.py roofline_syrk): per evaluation they are the launches of the 128x128-tile
# (C2: 64x64-tile) kernel with the largest grids -- 15 at C3, 5 at C2 (one per panel but the last)
nsyrk = {"c3": 15, "c2": 7, "c4": 15}.get(sys.argv[4])      # outer panels but the last (gpn_potrf_panel_width: 2048 | 1024 | 4096)
if nsyrk:
    import math
    match = "gemm_nt_kernel<"        # 128x128-tile launches and, for the last panels, 64x64-tile ones
    fg, wg = per_grid(sys.argv[1], "FETCH_SIZE", match), per_grid(sys.argv[2], "WRITE_SIZE", match)

    def triangular(w):          # lower-tile launches have mt (mt + 1) / 2 workgroups
        t = (math.isqrt(8 * w + 1) - 1) // 2
        # in-panel (rectangular) launches stay below ~1030 workgroups at C3/C4 and ~260 at C2: above that a
        # triangular grid is a SYRK trailing update (only the last, smallest one per evaluation falls under it)
        return t * (t + 1) // 2 == w and w > (300 if sys.argv[4] == "c2" else 1100)
    top = [g for g in fg if triangular(g)]
    nl = sum(fg[g][0] for g in top)
    if nl and all(g in wg for g in top):
        out["syrk_launches"] = nl
        out["syrk_bytes_per_launch"] = sum(fg[g][1] for g in top) * 2048.0 / nl + sum(wg[g][1] for g in top) * 1024.0 / sum(wg[g][0] for g in top)
        out["syrk_note"] = "lower-tile launches = triangular grids above the in-panel launch sizes (%d per evaluation minus the last, smallest one); FETCH x2 as above" % nsyrk
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
