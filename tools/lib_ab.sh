#!/bin/bash
# GPU-box tool: same-box A/B of two builds of libgpnative.so on whole evaluations (alternating processes).
#   tools/lib_ab.sh OLD.so [workloads...]     (NEW = the in-tree build; bench.py honours GPN_LIB)
OLD="$1"; shift
WLS="${@:-c2 c3}"
NEW="$(dirname "$0")/../gptorch_amd/lib/libgpnative.so"
for wl in $WLS; do
  steps=100; [ "$wl" != "c2" ] && steps=10
  for round in 1 2 3; do
    for tag in old new; do
      lib="$OLD"; [ "$tag" = new ] && lib="$NEW"
      ms=$(GPN_LIB="$lib" python bench.py --workload $wl --no-extras --no-cpu-baseline --no-fit --steps $steps --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms  lml %.12e' % (d['ms_per_step'], d.get('lml', float('nan'))))")
      echo "$wl round $round $tag: $ms"
    done
  done
done
for tag in old new; do
  lib="$OLD"; [ "$tag" = new ] && lib="$NEW"
  echo "leaf $tag: $(GPN_LIB="$lib" python tools/leaf_bench.py 2>/dev/null | tail -1)"
done
