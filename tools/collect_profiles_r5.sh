#!/bin/bash
# GPU-box script (round 5): regenerate the evidence under profiles/ for the current code state.
#   gpurun --timeout 4200 -- 'bash tools/collect_profiles_r5.sh r5'
# = everything collect_profiles_r4.sh collects (default bench line, kernel stats forward / backward at C2 and C3, PMC traffic,
# VALU-issue counters, MFMA busy, VFE, composite trace, predict, lock-step forward, leaf) plus the round-5 additions: kernel
# stats of the LOCK-STEP BACKWARD and of a whole lock-step step (C2 x 8, C1 x 64), the lock-step fit timings, the refinement's
# back-substitution A/B, and the A/Bs recorded as negatives (tile rule of batched K-clipped launches, left-looking inner
# panels, split assembly), the C5-shaped extended-precision check.
set -u
TAG=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
bash $R/tools/collect_profiles_r4.sh $TAG > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_bwd_c2 -o b8 -- \
    python3 $R/tools/fit_batched_profile.py c2 8 bwd > $O/stats_batched_bwd_c2.log 2>&1
T=$(ls $O/stats_batched_bwd_c2/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 $R/tools/trace_summary.py $T 6 > $O/trace_summary_batched_bwd_c2.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_step_c2 -o s8 -- \
    python3 $R/tools/fit_batched_profile.py c2 8 > $O/stats_batched_step_c2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_batched_step_c1 -o s64 -- \
    python3 $R/tools/fit_batched_profile.py c1 64 > $O/stats_batched_step_c1.log 2>&1
rm -f $O/stats_batched_*/*kernel_trace.csv $O/stats_batched_*/*agent_info.csv
cd $R
python3 tools/fit_batched_bench.py c2 1 2 4 8 16 --parts > $O/fit_batched_c2.txt 2>&1
python3 tools/fit_batched_bench.py c1 8 64 256 > $O/fit_batched_c1.txt 2>&1
python3 tools/backsub_ab.py > $O/backsub_ab.txt 2>&1
python3 tools/tri_tile_ab.py 8 > $O/tri_tile_ab.txt 2>&1
python3 tools/split_asm_ab.py > $O/split_asm_ab.txt 2>&1
python3 tools/vfe_extended_check.py > $O/vfe_extended_check.txt 2>&1
( python3 tools/outer_ab.py 32768 1 512:2048 512:2048:0x800000; python3 tools/outer_ab.py 16384 1 256:1024 256:1024:0x800000;
  python3 tools/outer_ab.py 8192 8 256:1024 256:1024:0x800000 ) > $O/inner_left_ab.txt 2>&1
ls -la $O | head -100
