#!/bin/bash
# developer helper: gpurun with a wait for a free slot (exit code 3 = no box/slot, nothing charged; every other code is final)
# usage: scripts/gpu.sh TIMEOUT 'command'
T=$1; shift
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 90
done
exit 3
