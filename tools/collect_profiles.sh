#!/bin/bash
# GPU-box script: regenerate the evidence under profiles/ for the current code state.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r1_final'
# Writes into gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards
# (tools/collect_profiles.sh does not touch profiles/ itself: gpurun only merges gpurun_out/).
set -u
TAG=${1:-r1_final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for WL in c2 c3; do
  STEPS=20; [ $WL = c3 ] && STEPS=5
  # 1. the bench line itself (with cpu baseline only for c2, the default workload)
  if [ $WL = c2 ]; then python3 $R/bench.py --workload $WL --steps $STEPS > $O/bench_$WL.json 2> $O/bench_$WL.err
  else python3 $R/bench.py --workload $WL --steps $STEPS --no-cpu-baseline > $O/bench_$WL.json 2> $O/bench_$WL.err; fi
  # 2. kernel trace + stats of the same command (no extras: forward evaluations only)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$WL -o $WL -- \
      python3 $R/bench.py --workload $WL --steps $STEPS --no-extras --no-cpu-baseline > $O/stats_$WL.log 2>&1
  # 3. HBM traffic counters, one pass each, counters only
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o f -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_fetch_$WL.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o w -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_write_$WL.log 2>&1
  F=$(ls $O/pmc_fetch_$WL/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $O/pmc_write_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 $R/tools/pmc_traffic.py $F $W $O/traffic_$WL.json $WL > /dev/null
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL           # raw counter dumps are large
  # 4. MFMA utilisation of the contraction launches (counters only, own pass)
  if [ $WL = c3 ]; then
    timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_$WL -o m -- \
        python3 $R/bench.py --workload $WL --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_mfma_$WL.log 2>&1
    M=$(ls $O/pmc_mfma_$WL/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$M" ] && python3 $R/tools/pmc_mfma.py $M $O/mfma_utilisation_$WL.json $WL > /dev/null
    rm -rf $O/pmc_mfma_$WL
  fi
  # keep only the stats summary + a compact per-grid trace summary
  T=$(ls $O/stats_$WL/*kernel_trace.csv 2>/dev/null | head -1)
  EV=$((3 + 2 * STEPS))   # warm-up + timed steps + event-profiled steps
  [ -n "$T" ] && python3 $R/tools/trace_summary.py $T $EV > $O/trace_summary_$WL.txt 2>&1
  rm -f $O/stats_$WL/*kernel_trace.csv $O/stats_$WL/*agent_info.csv
done
# config 5 (VFE): timing line + kernel stats
python3 $R/tools/vfe_bench.py > $O/vfe_c5.json 2> $O/vfe_c5.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c5 -- \
    python3 $R/tools/vfe_bench.py --steps 1 > $O/stats_c5.log 2>&1
rm -f $O/stats_c5/*kernel_trace.csv $O/stats_c5/*agent_info.csv
ls -la $O $O/stats_c2 | head -40
