#!/bin/bash
# A/B/C... on ONE box: alternates the given libraries (names under scripts/dev/, or "tree" = the in-tree libsss.so)
#   bash scripts/dev/abn.sh 3 "libsss_base.so tree libsss_exp4.so" 1024,1000000,128,10,f16 ...
REPS=$1; LIBS=$2; shift 2
for r in $(seq 1 $REPS); do
  for l in $LIBS; do
    echo "== $l"
    if [ "$l" = tree ]; then python3 scripts/quick_search_bench.py "$@" 2>/dev/null | cut -c1-160; else python3 scripts/dev/qb_lib.py $l "$@" 2>/dev/null | cut -c1-160; fi
  done
done
