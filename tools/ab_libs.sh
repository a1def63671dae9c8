#!/bin/bash
# A/B of device-library builds (e.g. -DSELL_UNROLL=8): alternate processes on the same GPU, two rounds.
# usage: tools/ab_libs.sh libA.so libB.so ...   ("default" = the in-tree library)
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset NGSAMG_HIP_LIB; else export NGSAMG_HIP_LIB="$lib"; fi
    echo "== $lib (round $round)"
    AB_INSTANCES=2 AB_ONLY=default python tools/ab_cycle.py 215 2>&1 | grep "median" || exit 1
  done
done
