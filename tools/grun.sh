#!/bin/bash
# tools/grun.sh <timeout> <script>: gpurun with retries while the pool has no free box / slot (exit code 3: nothing charged)
T=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
