#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md 'HBM' prescribes) of `bench.py --no-extras` into
HBM bytes per launch of the contraction kernel.  gfx950 corrections: the counters are in KB;
FETCH_SIZE reports exactly half of the bytes of wide (16 B/lane) coalesced reads -- the
kernel's operand traffic is LDS-DMA dwordx4, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <workload>"""
import collections, csv, json, sys

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
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
