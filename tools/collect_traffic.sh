#!/bin/bash
# GPU-box script: only the HBM-traffic PMC passes of tools/collect_profiles_r2.sh (FETCH_SIZE / WRITE_SIZE, separate runs).
#   gpurun --timeout 1200 -- 'bash tools/collect_traffic.sh r2t'
set -u
TAG=${1:-r2t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for WL in c3 c2; do
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -o f -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_fetch_$WL.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -o w -- \
      python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/pmc_write_$WL.log 2>&1
  F=$(ls $O/pmc_fetch_$WL/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $O/pmc_write_$WL/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && head -2 $F > $O/counter_csv_header_$WL.txt
  [ -n "$F" ] && [ -n "$W" ] && python3 $R/tools/pmc_traffic.py $F $W $O/traffic_$WL.json $WL
  rm -rf $O/pmc_fetch_$WL $O/pmc_write_$WL
done
