#!/bin/bash
# GPU-box tool: A/B of the grouped tile order's group height (tile rows per group; 8 in the product) on whole C3 evaluations --
# the tools' build reads GPN_GEMM_GROUP_H at first use.  Alternating processes on one box.
#   tools/group_h_ab.sh [heights...]
HS="${@:-8 4 16 32}"
DBG="$(dirname "$0")/../gptorch_amd/lib/libgpnative_dbg.so"
for round in 1 2; do
  for h in $HS; do
    ms=$(GPN_LIB="$DBG" GPN_GEMM_GROUP_H=$h python bench.py --workload c3 --no-extras --no-cpu-baseline --no-fit --steps 8 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms  syrk %.2f TFLOP/s  lml %.10f' % (d['ms_per_step'], d['roofline_syrk']['achieved'], d.get('lml', float('nan'))))")
    echo "c3 round $round group height $h: $ms"
  done
done
