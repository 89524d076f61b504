#!/bin/bash
# A/B on ONE box: alternates scripts/dev/libsss_base.so and the in-tree libsss.so over the given quick_search_bench shapes.
#   bash scripts/dev/ab.sh 3 1024,1000000,128,10,f16 ...        (3 alternations)
REPS=$1; shift
for r in $(seq 1 $REPS); do
  echo "== base";  python3 scripts/dev/qb_lib.py libsss_base.so "$@" 2>/dev/null | cut -c1-160
  echo "== new";   python3 scripts/quick_search_bench.py "$@" 2>/dev/null | cut -c1-160
done
