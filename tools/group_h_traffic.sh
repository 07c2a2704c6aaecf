#!/bin/bash
# GPU-box tool: fabric read traffic (FETCH_SIZE PMC pass, gfx950 corrections of tools/pmc_traffic.py) of C3's trailing updates for two
# group heights of the grouped tile order (tools' build, GPN_GEMM_GROUP_H).   gpurun -- 'bash tools/group_h_traffic.sh 8 4'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/group_h_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPN_LIB=$R/gptorch_amd/lib/libgpnative_dbg.so
for H in "$@"; do
  export GPN_GEMM_GROUP_H=$H
  rm -rf $O/f_$H $O/w_$H
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$H -o f -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-fit > $O/f_$H.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$H -o w -- python3 $R/bench.py --workload c3 --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-fit > $O/w_$H.log 2>&1
  F=$(ls $O/f_$H/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $O/w_$H/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 $R/tools/pmc_traffic.py $F $W $O/traffic_c3_group_h_$H.json c3 && echo "group height $H:" && python3 -c "
import json; d=json.load(open('$O/traffic_c3_group_h_$H.json')); print({k: d[k] for k in d if 'syrk' in k or 'per_launch' in k})"
  rm -rf $O/f_$H $O/w_$H
done
