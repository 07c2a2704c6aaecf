R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fs; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o fs -- python3 $R/tools/outer_ab.py 32768 1 0:0:0x100000 > $O/stats.log 2>&1
cd $R; head -12 $O/stats/fs_kernel_stats.csv | cut -c1-170
python3 tools/trace_timeline.py $O/stats/fs_kernel_trace.csv 2>/dev/null | head -5
python3 - <<'P'
import csv,sys,os
rows=list(csv.DictReader(open(os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/fs/stats/fs_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find a window in the last evaluation: print 60 consecutive kernels around a chain_step
idx=[i for i,r in enumerate(rows) if "chain_step" in r["Kernel_Name"]]
i0=idx[len(idx)-3000] if len(idx)>3000 else idx[0]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0+40]:
    print("%9.1f us + %7.1f us  q%s grid=%7s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Queue_Id","?"),r.get("Grid_Size","?"),r["Kernel_Name"][:60]))
P
rm -f $O/stats/*kernel_trace.csv $O/stats/*agent_info.csv
