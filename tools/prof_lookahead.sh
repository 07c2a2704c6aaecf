# rocprofv3 kernel trace of the look-ahead schedule (round 6): timeline windows as text under gpurun_out/r6
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6/trace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
N=${1:-8192}; shift
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/la -o la -- python3 $R/tools/lookahead_trace.py $N 1 "$@" > $O/la.log 2>&1
cd $R
F=$(find $O/la -name "la_kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $F persistent 7 260 > $R/gpurun_out/r6/lookahead_timeline_$N.txt 2>&1
rm -rf $O
