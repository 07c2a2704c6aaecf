#!/bin/bash
# GPU-box script (round 4): regenerate the evidence under profiles/ for the current code state.
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles_r4.sh r4'
# = everything collect_profiles_r3.sh collects (default bench line, kernel stats forward / backward at C2 and C3, PMC traffic,
# VALU-issue counters, MFMA busy, VFE, composite trace) plus the round-4 additions: kernel stats of GPR._predict (C2, C3),
# of a lock-step batch (C2 x 8, C1 x 64), the leaf's timeline and the three micro-benchmarks behind its design.
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
bash $R/tools/collect_profiles_r3.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for WL in c2 c3; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_predict_$WL -o predict_$WL -- \
      python3 $R/tools/predict_profile.py $WL blocked > $O/stats_predict_$WL.log 2>&1
  rm -f $O/stats_predict_$WL/*kernel_trace.csv $O/stats_predict_$WL/*agent_info.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_c2 -o b8 -- \
    python3 $R/tools/batched_profile.py c2 8 > $O/stats_batched_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_c1 -o b64 -- \
    python3 $R/tools/batched_profile.py c1 64 > $O/stats_batched_c1.log 2>&1
rm -f $O/stats_batched_c*/*kernel_trace.csv $O/stats_batched_c*/*agent_info.csv
cd $R
python3 tools/leaf16_check.py > $O/leaf16_check.txt 2>&1
python3 tools/batched_bench.py c2 1 2 4 8 16 > $O/batched_c2.txt 2>&1
python3 tools/batched_bench.py c1 8 64 256 > $O/batched_c1.txt 2>&1
python3 tools/predict_bench.py c2 > $O/predict_c2.txt 2>&1
python3 tools/predict_bench.py c3 > $O/predict_c3.txt 2>&1
[ -x tools/bin/lat_bench ] && ./tools/bin/lat_bench > $O/lat_bench.txt 2>&1
[ -x tools/bin/prio_bench ] && ./tools/bin/prio_bench > $O/prio_bench.txt 2>&1
ls -la $O | head -80
