#!/bin/bash
# GPU-box script (round-3 review item 8): 256x128 macro tile vs the shipped 128x128 tile on C3's first trailing update --
# same-box timing (tools/gemm_ab.py, interleaved rounds, results compared) and HBM-side traffic (separate --pmc passes).
#   gpurun --timeout 1500 -- 'bash tools/macro_tile_ab.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/macro; mkdir -p $O
cd $R
GPN_AB_ROUNDS=5 timeout 900 python3 tools/gemm_ab.py 11,12,13 30720 30720 2048 1 8192 8192 8192 0 16384 16384 4096 1 8448 8448 2048 1 2>&1 | grep -v amdgpu.ids > $O/gemm_ab.txt
cat $O/gemm_ab.txt
cd /tmp && export TMPDIR=/tmp
for v in 11 12 13; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_v$v -o f -- python3 $R/tools/macro_tile_run.py $v > $O/pmc_fetch_v$v.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_v$v -o w -- python3 $R/tools/macro_tile_run.py $v > $O/pmc_write_v$v.log 2>&1
done
cd $R
python3 tools/macro_tile_traffic.py $O $O/traffic_c3_macro.json
find $O -name "*counter_collection.csv" -size +2M -delete
