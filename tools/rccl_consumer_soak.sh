#!/bin/bash
# GPU-box tool: run examples/dist_consumer.c N times (1 x 1 grid, collectives forced through RCCL) and report how
# long each run takes / whether one hangs in the RCCL bootstrap (NCCL_DEBUG=INFO of a timed-out run is kept).
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-12}
mkdir -p $R/build $R/gpurun_out
gcc -std=c99 $R/examples/dist_consumer.c -I$R/include -I/opt/rocm/include -L$R/gptorch_amd/lib -lgpnative -lgpnative_rccl \
    -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$R/gptorch_amd/lib -Wl,-rpath,/opt/rocm/lib -lm -o $R/build/dist_consumer || exit 1
for i in $(seq 1 $N); do
  t0=$(date +%s.%N)
  NCCL_DEBUG=INFO timeout 90 $R/build/dist_consumer 2048 8 512 > /tmp/dc_$i.log 2>&1
  rc=$?
  t1=$(date +%s.%N)
  echo "run $i rc=$rc $(python3 -c "print(round($t1 - $t0, 1))") s  $(grep -o 'lml=.*' /tmp/dc_$i.log | head -1)"
  if [ $rc -ne 0 ]; then cp /tmp/dc_$i.log $R/gpurun_out/dist_consumer_failed_$i.log; fi
done
