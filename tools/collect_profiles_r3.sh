#!/bin/bash
# GPU-box script (round 3): regenerate the evidence under profiles/ for the current code state.
#   gpurun --timeout 3000 -- 'bash tools/collect_profiles_r3.sh r3'
# = everything collect_profiles_r2.sh collects (default bench line incl. the measured full-size CPU baseline and the 50-step
# fit, kernel stats of forward / backward at C2 and C3, PMC traffic, VALU-issue counters, MFMA busy, VFE) plus the
# round-3 additions: the composite-kernel trace and the refinement's kernel stats (inside the C3 forward stats).
set -u
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
bash $R/tools/collect_profiles_r2.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_composite -o composite -- \
    python3 $R/tools/composite_profile.py 8192 > $O/stats_composite.log 2>&1
rm -f $O/stats_composite/*kernel_trace.csv $O/stats_composite/*agent_info.csv
ls -la $O | head -80
