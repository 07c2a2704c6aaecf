#!/bin/bash
# GPU-box script (round 6): regenerate the evidence under profiles/ for the current code state (self-contained: the scripts of
# rounds 2-5 are in the history).   gpurun --timeout 3000 -- 'bash tools/collect_profiles_r6.sh r6'
# Writes into gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
#   0  the default bench line exactly as the driver runs it
#   1  kernel trace + stats of the headline leg (C3) and of C2, forward evaluations only
#   2  the backward (loss() + backward()): kernel stats at C2 and C3
#   3  HBM traffic counters (one --pmc pass each; FETCH / WRITE as MI355X_MICROARCH.md prescribes) -> traffic_c{2,3}.json
#   4  MFMA-busy of the contraction launches at C3, VALU issue of the HBM-side kernels
#   5  round 6: the persistent factorisation against the launch-based driver (sizes, per-task trace at C2), the optimiser step
#      as one graph replay, the lock-step fit timings, sparse (VFE) restarts in lock step
set -u
TAG=${1:-r6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
for WL in c3 c2; do
  STEPS=5; [ $WL = c2 ] && STEPS=20
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$WL -o $WL -- \
      python3 $R/bench.py --workload $WL --steps $STEPS --no-extras --no-cpu-baseline > $O/stats_$WL.log 2>&1
  T=$(ls $O/stats_$WL/*kernel_trace.csv 2>/dev/null | head -1)
  EV=$((3 + 2 * STEPS))
  [ -n "$T" ] && python3 $R/tools/trace_summary.py $T $EV > $O/trace_summary_$WL.txt 2>&1
  rm -f $O/stats_$WL/*kernel_trace.csv $O/stats_$WL/*agent_info.csv
  BS=4; [ $WL = c3 ] && BS=2
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bwd_$WL -o bwd_$WL -- \
      python3 $R/tools/backward_profile.py $WL $BS > $O/stats_bwd_$WL.log 2>&1
  rm -f $O/stats_bwd_$WL/*kernel_trace.csv $O/stats_bwd_$WL/*agent_info.csv
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o f -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_fetch_$WL.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o w -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_write_$WL.log 2>&1
  F=$(ls $O/pmc_fetch_$WL/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $O/pmc_write_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 $R/tools/pmc_traffic.py $F $W $O/traffic_$WL.json $WL > /dev/null
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu_$WL -o v -- \
      python3 $R/tools/backward_profile.py $WL 1 > $O/pmc_valu_$WL.log 2>&1
  V=$(ls $O/pmc_valu_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$V" ] && python3 $R/tools/pmc_valu.py $V $O/valu_bound_$WL.json $WL > /dev/null
  rm -rf $O/pmc_valu_$WL
done
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma_c3 -o m -- \
    python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_mfma_c3.log 2>&1
M=$(ls $O/pmc_mfma_c3/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$M" ] && python3 $R/tools/pmc_mfma.py $M $O/mfma_utilisation_c3.json c3 > /dev/null
rm -rf $O/pmc_mfma_c3
cd $R
cd /tmp
timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_valu_c4 -o v -- \
    python3 $R/tools/backward_profile.py c4 1 > $O/pmc_valu_c4.log 2>&1
V=$(ls $O/pmc_valu_c4/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$V" ] && python3 $R/tools/pmc_valu.py $V $O/valu_bound_c4.json c4 > /dev/null
rm -rf $O/pmc_valu_c4
cd $R
python3 tools/persistent_ab.py 4096 6144 8192 12288 16384 --reps 4 > $O/persistent_ab.txt 2>&1
PP_PATHS=2 PP_DEPTH=14 python3 tools/persistent_trace.py 8192 16 > $O/persistent_trace_c2.txt 2>&1
python3 tools/capture_bench.py 512 1024 2048 4096 8192 --iters 100 > $O/capture_bench.txt 2>&1
python3 tools/fit_batched_bench.py c2 1 8 --parts > $O/fit_batched_c2.txt 2>&1
python3 tools/kmat_bench.py c2 c3 c4 > $O/kmat_bench.txt 2>&1
python3 tools/vfe_batched_bench.py --parts > $O/vfe_batched_bench.txt 2>&1
python3 tools/vfe_fit_probe.py > $O/vfe_fit_probe.txt 2>&1
python3 tools/fit_batched_bench.py c1 8 64 --capture --steps 400 > $O/capture_lockstep.txt 2>&1
ls -la $O | head -60
