"""Aggregate a rocprofv3 kernel_trace.csv of tools/vfe_fwd_once.py by (kernel, workgroup count) for the
SECOND evaluation in the trace.  usage: trace_by_grid.py <rocprof output dir>"""
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
kidx=[i for i,r in enumerate(rows) if "kmat" in r["Kernel_Name"]]
half=kidx[len(kidx)//2]
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows[half-2:]:
    k=r["Kernel_Name"].replace("void gpn::","")[:46]+" wg="+str(int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]))
    a=agg[k]; a[0]+=1; a[1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
tot=sum(v[1] for v in agg.values())
print("second forward: span %.1f ms, kernel sum %.1f ms" % ((int(rows[-1]["End_Timestamp"])-int(rows[half-2]["Start_Timestamp"]))/1e6, tot/1e3))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:10]:
    print("%-64s n=%4d total %9.1f us avg %8.1f" % (k, v[0], v[1], v[1]/v[0]))
