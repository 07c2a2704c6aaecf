#!/bin/bash
# GPU-box script (round 2): regenerate the evidence under profiles/ for the current code state.
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles_r2.sh r2'
# Writes into gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 0. the default bench line exactly as the driver runs it (headline C3 + keyed extras + cpu baseline)
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
for WL in c3 c2; do
  STEPS=5; [ $WL = c2 ] && STEPS=20
  # 1. kernel trace + stats of the headline leg (forward evaluations only)
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$WL -o $WL -- \
      python3 $R/bench.py --workload $WL --steps $STEPS --no-extras --no-cpu-baseline > $O/stats_$WL.log 2>&1
  T=$(ls $O/stats_$WL/*kernel_trace.csv 2>/dev/null | head -1)
  EV=$((3 + 2 * STEPS))
  [ -n "$T" ] && python3 $R/tools/trace_summary.py $T $EV > $O/trace_summary_$WL.txt 2>&1
  rm -f $O/stats_$WL/*kernel_trace.csv $O/stats_$WL/*agent_info.csv
  # 2. the backward (loss()+backward() steps): kernel stats
  BS=4; [ $WL = c3 ] && BS=2
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_$WL -o bwd_$WL -- \
      python3 $R/tools/backward_profile.py $WL $BS > $O/stats_bwd_$WL.log 2>&1
  T=$(ls $O/stats_bwd_$WL/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$T" ] && python3 $R/tools/trace_summary.py $T $((BS + 1)) > $O/trace_summary_bwd_$WL.txt 2>&1
  rm -f $O/stats_bwd_$WL/*kernel_trace.csv $O/stats_bwd_$WL/*agent_info.csv
  # 3. HBM traffic counters, one pass each, counters only
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o f -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_fetch_$WL.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o w -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_write_$WL.log 2>&1
  F=$(ls $O/pmc_fetch_$WL/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $O/pmc_write_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 $R/tools/pmc_traffic.py $F $W $O/traffic_$WL.json $WL > /dev/null
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL
  # 4. VALU issue counters of the HBM-side kernels (K assembly in the forward, gradient sweep in the backward)
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu_$WL -o v -- \
      python3 $R/tools/backward_profile.py $WL 1 > $O/pmc_valu_$WL.log 2>&1
  V=$(ls $O/pmc_valu_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$V" ] && python3 $R/tools/pmc_valu.py $V $O/valu_bound_$WL.json $WL > /dev/null
  rm -rf $O/pmc_valu_$WL
done
# 5. MFMA utilisation of the contraction launches at C3 (counters only, own pass)
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c3 -o m -- \
    python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_mfma_c3.log 2>&1
M=$(ls $O/pmc_mfma_c3/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$M" ] && python3 $R/tools/pmc_mfma.py $M $O/mfma_utilisation_c3.json c3 > /dev/null
rm -rf $O/pmc_mfma_c3
# 6. config 5 (VFE): kernel stats of one bound evaluation
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -o c5 -- \
    python3 $R/tools/vfe_bench.py --steps 1 > $O/stats_c5.log 2>&1
rm -f $O/stats_c5/*kernel_trace.csv $O/stats_c5/*agent_info.csv
ls -la $O | head -60
