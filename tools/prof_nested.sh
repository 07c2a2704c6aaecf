R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nested; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_c2 -o b8 -- python3 $R/tools/batched_profile.py c2 8 > $O/stats_batched_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -o c3 -- python3 $R/bench.py --workload c3 --no-extras --no-cpu-baseline > $O/stats_c3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c2 -o c2 -- python3 $R/bench.py --workload c2 --no-extras --no-cpu-baseline > $O/stats_c2.log 2>&1
cd $R; python3 tools/trace_summary.py $O/stats_c3/c3_kernel_trace.csv > $O/trace_summary_c3.txt 2>&1; python3 tools/trace_summary.py $O/stats_batched_c2/b8_kernel_trace.csv > $O/trace_summary_b8.txt 2>&1
python3 tools/trace_summary.py $O/stats_c2/c2_kernel_trace.csv > $O/trace_summary_c2.txt 2>&1
rm -f $O/*/*kernel_trace.csv $O/*/*agent_info.csv
head -12 $O/stats_batched_c2/b8_kernel_stats.csv | cut -c1-160
