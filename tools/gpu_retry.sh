#!/bin/bash
# tools/gpu_retry.sh <timeout-seconds> '<command>': gpurun, retried every 2 minutes while no slot / box is free (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
